# SQ counters of csrc/conv1x1.hip's kernels on the 64 -> 128 layer (tools/bench_conv1x1.py, ONLY=3): where the waves' cycles go.
# usage (GPU box, repo root): bash tools/probes/pmc_conv1x1_sq.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc2; mkdir -p /tmp/pc2
i=0
for grp in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ONLY=${ONLY:-3} timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pc2/g$i -o p -- python $R/tools/bench_conv1x1.py > /tmp/pc2/log$i 2>&1 < /dev/null || tail -3 /tmp/pc2/log$i
done
python - $(find /tmp/pc2 -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        fd = 'conv1x1_kernel' in n or 'conv1x1_rows_kernel' in n or 'conv1x1_split_kernel' in n
        key = ('forward' if fd and ', false' in n else 'data_gradient' if fd
               else 'weight_gradient' if 'wgrad1x1_ball_kernel' in n else None)
        if key:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
