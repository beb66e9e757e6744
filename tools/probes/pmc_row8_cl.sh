# r06: project_rows_kernel with and without the channels-last copies (hcm_project_rows_cl) under the L1 / L2 request counters and the
# wait counters, separate --pmc passes, no trace domains beside --kernel-trace.  usage: pmc_row8_cl.sh  (writes to stdout)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cl in 0 1; do
  echo "== ROW8_CL=$cl"
  ROW8_CL=$cl python $R/tools/probes/row8_probe.py 256 32 20 2>&1 | grep -i "project\|forward" | head -4
  for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/pr
    ROW8_CL=$cl rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pr -o p -- python $R/tools/probes/row8_probe.py 256 32 3 > /tmp/pr.log 2>&1 || tail -3 /tmp/pr.log
    python - "project_rows_kernel,nchw_to_nhwc_kernel" $(find /tmp/pr -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        for k in sys.argv[1].split(','):
            if k in r.get('Kernel_Name', ''):
                acc[k, r['Counter_Name']].append(float(r['Counter_Value']))
for k in sys.argv[1].split(','):
    d = {c: (round(sum(v) / len(v)), len(v)) for (kk, c), v in sorted(acc.items()) if kk == k}
    if d:
        print(k, d)
PY
  done
done
