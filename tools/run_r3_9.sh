set -x
O=gpurun_out/r3_9; mkdir -p $O
(timeout 2400 python -m pytest tests/test_exact_gpu.py tests/test_encoder_bf16_gpu.py tests/test_trace_moco.py tests/test_bank_gpu.py -x -q -m gpu -s 2>&1 | grep -v "^\[\|Gloo\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25) > $O/pytest.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1) > $O/bench_default.json
bash tools/run_bank_sweep.sh $O/bank_sweep > $O/bank_sweep.log 2>&1
tail -n 14 $O/pytest.log; python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'], d.get('checked')); print(d['roofline_secondary']); print(d['cpu_baseline'])"; cat $O/bank_sweep/pmc.jsonl; cat $O/bank_sweep/in_step.jsonl; python -c "
import json
for r in json.load(open('$O/bank_sweep/time.json')): print(r['n_data'], r['K'], r['dtype'], r['variant'], r.get('back_to_back_GBps'), r.get('cold_GBps'))"
