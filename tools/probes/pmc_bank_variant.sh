# FETCH_SIZE / WRITE_SIZE of the bank pass (separate --pmc passes).  usage: pmc_bank_variant.sh [n_data] [K] [dtype]
# (r04 took a kernel variant as first argument; the variants were removed in r05)
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1048576}; K=${2:-16384}; DT=${3:-fp32}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcv; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcv -- python $R/tools/bank_sweep.py pmc $N $K $DT > /dev/null 2>&1
  python - $c $(find /tmp/pmcv -name '*counter_collection.csv' | head -1) <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if r['Counter_Name'] == sys.argv[1]:
        acc[r['Kernel_Name'].split('(')[0][-50:]].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:4]:
    print('  %-12s %-52s launches %3d  mean %.0f KiB' % (sys.argv[1], k, len(v), sum(v) / len(v)))
PY
done
