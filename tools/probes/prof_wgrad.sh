# GPU time of hcm_conv3x3_wgrad stand-alone (tools/pmc_wgrad.py) under rocprofv3, SHAPE=N,C,H
cd /tmp && export TMPDIR=/tmp
for shp in 32,18,64 32,36,32; do
rm -rf /tmp/wg; SHAPE=$shp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg -- python $GRAFT_REPO_ROOT/tools/pmc_wgrad.py > /tmp/wg.log 2>&1; f=$(find /tmp/wg -name "*kernel_stats.csv" | head -1); python -c "
import csv
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'wgrad' in n: print('$shp', n[28:90].ljust(62), r['Calls'], round(float(r['AverageNs'])/1e3,2))
"
done
