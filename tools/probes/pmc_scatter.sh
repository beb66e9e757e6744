# Where the cycles of scatter_planned_kernel go on the pts2depth backward: SQ counters, two --pmc passes per index pattern
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in random nn hub; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/ps
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/ps -o p -- python $R/tools/probes/scatter_case.py $mode > /tmp/ps.log 2>&1 || tail -3 /tmp/ps.log
  grep " ms" /tmp/ps.log
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
