# Memory side of scatter_planned_kernel on the pts2depth backward: fabric bytes and L2 hit rate
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in hub random; do
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  rm -rf /tmp/ps
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ps -o p -- python $R/tools/probes/scatter_case.py $mode > /tmp/ps.log 2>&1 || tail -3 /tmp/ps.log
  python - $(find /tmp/ps -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'scatter_planned_kernel' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
done
done
