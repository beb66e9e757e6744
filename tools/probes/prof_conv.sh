cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/cv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cv -- python $GRAFT_REPO_ROOT/tools/bench_conv.py $GRAFT_REPO_ROOT/hcmoco_amd/csrc/libhcmoco_hip.so > /tmp/cv.log 2>&1; f=$(find /tmp/cv -name "*kernel_stats.csv" | head -1); python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'conv3x3' in n: print(n[28:80], r['Calls'], round(float(r['AverageNs'])/1e3,2))
"
