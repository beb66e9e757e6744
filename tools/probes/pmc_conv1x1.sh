# HBM-side traffic of csrc/conv1x1.hip's three kernels on one ball-tensor layer (64 -> 128 channels, 1024 x 32 positions, B = 32:
# x 268 MB, z 537 MB), FETCH_SIZE and WRITE_SIZE in separate --pmc passes (tools/bench_conv1x1.py, ONLY=3).
# usage (GPU box, repo root): bash tools/probes/pmc_conv1x1.sh <tag>   -> gpurun_out/<tag>_conv1x1_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rXX}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc1; mkdir -p /tmp/pc1
for ctr in FETCH_SIZE WRITE_SIZE; do
  ONLY=${ONLY:-3} timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pc1/$ctr -o p -- python $R/tools/bench_conv1x1.py > /tmp/pc1/log_$ctr 2>&1 < /dev/null || tail -3 /tmp/pc1/log_$ctr
done
python - $R/gpurun_out/${TAG}_conv1x1_pmc.json $(find /tmp/pc1 -name '*counter_collection.csv') <<'PY'
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        n = r.get('Kernel_Name', '')
        fd = 'conv1x1_kernel' in n or 'conv1x1_rows_kernel' in n or 'conv1x1_split_kernel' in n
        key = ('forward' if fd and ', false' in n else 'data_gradient' if fd
               else 'weight_gradient' if 'wgrad1x1_ball_kernel' in n else None)
        if key:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
import os
B, C, K, P = {'1': (32, 32, 64, 131072), '3': (32, 64, 128, 32768), '4': (32, 128, 256, 8192)}[os.environ.get('ONLY', '3')]
x, z = 4 * B * C * P, 4 * B * K * P
alg = {'forward': (x, z), 'data_gradient': (z, x), 'weight_gradient': (x + z, 0)}
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    out[k] = {'FETCH_SIZE_KiB_mean': m.get('FETCH_SIZE'), 'WRITE_SIZE_KiB_mean': m.get('WRITE_SIZE'), 'launches': len(d.get('FETCH_SIZE', [])),
              'algorithmic_read_bytes': alg[k][0], 'algorithmic_write_bytes': alg[k][1],
              'read_ratio_raw': m.get('FETCH_SIZE', 0) * 1024 / alg[k][0],
              'write_ratio_raw': (m.get('WRITE_SIZE', 0) * 1024 / alg[k][1]) if alg[k][1] else None}
out['note'] = ('mean per launch; raw counter values.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide '
               '(16 B / lane) coalesced streaming read -- the weight gradient reads that way (expected raw ratio 0.5 = every byte '
               'once); the forward / data-gradient kernels read 4 B / lane in 64-byte pieces (uncalibrated width), and the second '
               'output-channel block of the forward re-reads x out of L2 / the Infinity Cache')
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
