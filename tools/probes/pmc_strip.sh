# Where the cycles of the strip kernels go: SQ counters of the stand-alone launches (tools/pmc_fmap.py), two --pmc passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SALU"; do
  rm -rf /tmp/ps
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ps -o p -- python $R/tools/pmc_fmap.py > /tmp/ps.log 2>&1 || tail -3 /tmp/ps.log
  python - $(find /tmp/ps -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if 'strip_kernel' in n and 'DensePolicy' in n:
            key = 'dense_grad' if 'DensePolicy, true' in n else 'dense_stats'
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
done
