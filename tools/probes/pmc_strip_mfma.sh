# MFMA counters of the dense / SCL strip kernels (stand-alone launches, tools/pmc_fmap.py): matrix ops issued by type, matrix-core
# busy cycles, wave cycles -- one --pmc pass per counter group, for the fp32-accurate (split-bf16) and the one-term bf16 kernels.
# usage (GPU box, repo root): bash tools/probes/pmc_strip_mfma.sh <tag>   -> gpurun_out/<tag>_strip_mfma_pmc_{fp32,bf16}.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for DT in fp32 bf16; do
  rm -rf /tmp/psm; mkdir -p /tmp/psm
  i=0
  for grp in "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    FMAP_DTYPE=$DT timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/psm/g$i -o p -- python $R/tools/pmc_fmap.py > /tmp/psm/log$i 2>&1 < /dev/null || tail -3 /tmp/psm/log$i
  done
  python - $R/gpurun_out/${TAG}_strip_mfma_pmc_$DT.json $(find /tmp/psm -name '*counter_collection.csv') <<'PY'
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if 'strip_kernel' not in n:
            continue
        pol = 'dense' if 'DensePolicy' in n else 'scl'
        key = pol + ('_grad' if 'Policy, true' in n else '_stats')
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in sorted(acc.items())}
out['note'] = ('mean per launch over the launches of tools/pmc_fmap.py (B = 32, S = 400, J = 17, 24 images with depth); '
               'SQ_INSTS_VALU_MFMA_MOPS_*: matrix operations issued, in units of 512 flop')
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for k, d in out.items():
    if isinstance(d, dict):
        print(k, {c: round(v) for c, v in d.items()})
PY
done
