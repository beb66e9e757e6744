set -x
O=gpurun_out/r3_23; mkdir -p $O
(timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -6) > $O/pytest_full.log 2>&1
tail -3 $O/pytest_full.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok") 
timeout 2400 bash tools/final_profiles.sh r03 > $O/final_profiles.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/r03_final_bench_line.json').read())
print(d['value'], d['ms_per_step'], d.get('checked'), d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['traffic_source'], d['config'].get('cpu_affinity'), d['cpu_baseline']['value'])
for l in open('gpurun_out/final/r03_secondary_configs.log'):
    if l.startswith('=='): print(l.strip()); continue
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['value'], d['ms_per_step'], d.get('checked'), d['roofline']['avg_launch_ms'], d['roofline']['achieved'])
for f in ('deterministic_mode','two_ranks_one_gpu_gloo','forced_collectives'):
    ls=[l for l in open('gpurun_out/final/r03_%s_bench_line.json'%f) if l.startswith('{')]
    print(f, (lambda d:(d['value'], d['ms_per_step'], d.get('checked'), d['n_gpus']))(json.loads(ls[-1])) if ls else 'no line')
PY
cat gpurun_out/final/r03_phase_times.txt | tail -10
