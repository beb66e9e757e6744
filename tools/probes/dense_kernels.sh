# Average durations of the dense-loss kernels (prep / table, stats, grad) from a kernel trace of a short bench run.
# usage: dense_kernels.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk -- python $R/bench.py --steps 10 --warmup 5 --no_cpu_baseline --no_check > /dev/null 2>&1
python - $(find /tmp/dk -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'dense_prep' in n or ('strip_kernel' in n and 'DensePolicy' in n) or 'dense_finish' in n or 'gather_norm' in n:
        print('  %-70s calls %4s  avg %8.2f us' % (n.split('(')[0][-70:], r['Calls'], float(r['AverageNs']) / 1e3))
        tot += float(r['AverageNs']) / 1e3
print('  sum of the averages: %.1f us' % tot)
PY
