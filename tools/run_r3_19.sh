set -x
O=gpurun_out/r3_19; mkdir -p $O
(timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -15) > $O/pytest_full.log 2>&1
tail -5 $O/pytest_full.log
timeout 2400 bash tools/final_profiles.sh r03 > $O/final_profiles.log 2>&1
tail -5 $O/final_profiles.log
cat gpurun_out/final/r03_final_bench_line.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('checked'), d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
cat gpurun_out/final/r03_secondary_configs.log | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('=='): print(l.strip()); continue
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['value'], d['ms_per_step'], d.get('checked'), d['roofline']['avg_launch_ms'], d['roofline']['achieved'])"
for f in deterministic_mode two_ranks_one_gpu_gloo forced_collectives; do python -c "
import json
ls=[l for l in open('gpurun_out/final/r03_${f}_bench_line.json') if l.startswith('{')]
print('$f', (lambda d:(d['value'], d['ms_per_step'], d.get('checked')))(json.loads(ls[-1])) if ls else 'no line')"; done
cat gpurun_out/final/r03_phase_times.txt
