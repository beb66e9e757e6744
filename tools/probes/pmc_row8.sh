# Row 8's kernels stand-alone (tools/probes/row8_probe.py): hipEvent timings, then SQ / TCP / TCC counters of the named
# kernel in separate --pmc passes (no trace domains beside --kernel-trace).  usage: pmc_row8.sh <kernel substring> [crop] [B]
R=$GRAFT_REPO_ROOT
K=${1:-branch_grad_kernel}
CROP=${2:-256}
BB=${3:-32}
cd /tmp && export TMPDIR=/tmp
[ -n "$SKIP_TIMING" ] || python $R/tools/probes/row8_probe.py $CROP $BB 20
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pr
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pr -o p -- python $R/tools/probes/row8_probe.py $CROP $BB 3 > /tmp/pr.log 2>&1 || tail -3 /tmp/pr.log
  python - "$K" $(find /tmp/pr -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        for k in sys.argv[1].split(','):
            if k in r.get('Kernel_Name', ''):
                acc[k, r['Counter_Name']].append(float(r['Counter_Value']))
for k in sys.argv[1].split(','):
    print(k, {c: (round(sum(v) / len(v)), len(v)) for (kk, c), v in sorted(acc.items()) if kk == k})
PY
done
