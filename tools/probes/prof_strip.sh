# GPU time of the strip kernels stand-alone (tools/pmc_fmap.py) under rocprofv3; HCM_LIB selects a diagnostic build
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/st; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $GRAFT_REPO_ROOT/tools/pmc_fmap.py > /tmp/st.log 2>&1; f=$(find /tmp/st -name "*kernel_stats.csv" | head -1); python -c "
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'strip_kernel' in n: print(n[28:90].ljust(62), r['Calls'], round(float(r['AverageNs'])/1e3,2))
"
