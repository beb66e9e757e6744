#!/bin/bash
# profiles/r04_bank_pass_sweep.json: timing sweep (tools/bank_sweep.py) + FETCH_SIZE / WRITE_SIZE of the default kernel
# per configuration (separate --pmc passes, gfx950 read correction: /opt/skills/guides/MI355X_MICROARCH.md) + the
# in-step duration from bench.py.   usage (GPU box, repo root): bash tools/run_bank_sweep.sh OUTDIR
set -x
R=$PWD; O=$R/${1:-gpurun_out/bank_sweep}; mkdir -p $O
python tools/bank_sweep.py time $O/time.json > $O/time.log 2>&1
cd /tmp && export TMPDIR=/tmp
: > $O/pmc.jsonl
for n in 131072 1048576 4194304; do for K in 16384 65536; do for dt in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcb; timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcb -- python $R/tools/bank_sweep.py pmc $n $K $dt > /dev/null 2>&1
    cp $(find /tmp/pmcb -name '*counter_collection.csv' | head -1) /tmp/pmc_$c.csv
  done
  python - $n $K $dt >> $O/pmc.jsonl <<'PY'
import csv, json, sys
n, K, dt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def mean(path, counter):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path)) if 'bank_pass' in r.get('Kernel_Name', '') and r['Counter_Name'] == counter]
    return (sum(v) / len(v), len(v)) if v else (float('nan'), 0)
f, nf = mean('/tmp/pmc_FETCH_SIZE.csv', 'FETCH_SIZE'); w, nw = mean('/tmp/pmc_WRITE_SIZE.csv', 'WRITE_SIZE')
alg = 32 * (3 * (K + 1) * 128 * (2 if dt == 'bf16' else 4) + (K + 1) * 8 + 12 * 128 * 4)
print(json.dumps({'n_data': n, 'K': K, 'dtype': dt, 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches': min(nf, nw),
                  'traffic_bytes': int(2 * f * 1024 + w * 1024), 'algorithmic_bytes': alg,
                  'traffic_over_algorithmic': round((2 * f * 1024 + w * 1024) / alg, 4)}))
PY
done; done; done
cd $R
: > $O/in_step.jsonl
for n in 131072 1048576; do for K in 16384 65536; do for dt in fp32 bf16; do
  python bench.py --steps 12 --warmup 4 --no_cpu_baseline --no_check --n_data $n --nce_k $K --bank_dtype $dt 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'n_data': $n, 'K': $K, 'dtype': '$dt', 'in_step_us': round(1e3*r['avg_launch_ms'],2), 'in_step_GBps': r['achieved'], 'samples_per_s': d['value']}))" >> $O/in_step.jsonl
done; done; done
python - $O <<'PY'
import json, sys, os
O = sys.argv[1]
rows = json.load(open(os.path.join(O, 'time.json')))
pmc = [json.loads(l) for l in open(os.path.join(O, 'pmc.jsonl')) if l.strip()]
ins = [json.loads(l) for l in open(os.path.join(O, 'in_step.jsonl')) if l.strip()]
json.dump({'peak_GBps': 8000, 'mall_bytes': 256 * 2 ** 20,
           'note': 'time: pass kernel only, hipEvents (hcm_prof); cold = a 1 GiB fill between launches (banks evicted from the '
                   'Infinity Cache, as between two training steps); pmc: default kernel variant, FETCH_SIZE and WRITE_SIZE in '
                   'separate rocprofv3 --pmc passes, read bytes = 2 x FETCH_SIZE x 1024 (gfx950 correction), back-to-back launches; '
                   'in_step: bench.py roofline (hipEvents inside the timed training steps)',
           'time': rows, 'pmc': pmc, 'in_step': ins}, open(os.path.join(O, 'r04_bank_pass_sweep.json'), 'w'), indent=1)
PY
