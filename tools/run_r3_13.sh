set -x
O=gpurun_out/r3_13; mkdir -p $O
bash tools/probes/pmc_scatter.sh > $O/scatter_sq_counters.txt 2>&1
cat $O/scatter_sq_counters.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_check --arch HRNetPN > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - $(find /tmp/kt -name '*kernel_stats.csv') <<'PY' > $O/hrnetpn_kernel_stats_top.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot / 1e6)
for r in rows[:45]:
    print('%-110s %6s calls %9.3f ms  avg %9.1f us' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print('--- point ops')
for r in rows:
    if any(k in r['Name'] for k in ('scatter', 'plan_kernel', 'three_', 'ball_', 'fps', 'group_', 'gather_points', 'rowmax', 'furthest')):
        print('%-110s %6s calls %9.3f ms  avg %9.1f us' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
cat $O/hrnetpn_kernel_stats_top.txt
