# per-kernel durations of the row-8 probe (rocprofv3 --kernel-trace --stats).  usage: prof_row8.sh [crop] [B]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr8 -o p -- python $R/tools/probes/row8_probe.py ${1:-256} ${2:-32} 10 > /tmp/pr8.log 2>&1 || tail -5 /tmp/pr8.log
python - $(find /tmp/pr8 -name '*kernel_stats.csv') <<'PY'
import csv, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        n = r['Name']
        if any(k in n for k in ('project_rows', 'proj_dw', 'stencil_plan', 'branch_grad', 'sample_branches', 'Cijk')):
            print('%-70s calls %5s  avg %9.1f us  min %9.1f  max %9.1f' % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
