#!/bin/bash
# MFMA counters of the strip kernels (dense soft-InfoNCE, SCL) at the bench size, fp32 and bf16 contractions.
# One --pmc group per pass (SQ has 8 slots); writes gpurun_out/pmc/strip_mfma_<dtype>.csv summaries.
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rm -rf /tmp/pmc_fmap_$dt
  FMAP_DTYPE=$dt rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d /tmp/pmc_fmap_$dt -o p -- python $R/tools/pmc_fmap.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc/strip_mfma_$dt.json strip_kernel $(find /tmp/pmc_fmap_$dt -name '*counter_collection.csv') > /dev/null
  python - "$R/gpurun_out/pmc/strip_mfma_${dt}_by_kernel.json" $(find /tmp/pmc_fmap_$dt -name '*counter_collection.csv') <<'PY'
import csv, json, sys
from collections import defaultdict
out, files = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        if 'strip_kernel' in n:
            key = ('dense' if 'DensePolicy' in n else 'scl') + ('_grad' if ', true,' in n.split('>(')[0] else '_stats')
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
PY
done
