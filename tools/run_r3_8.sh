set -x
O=gpurun_out/r3_8; mkdir -p $O
(timeout 1500 python -m pytest tests/test_section_gpu.py tests/test_fmap_gpu.py tests/test_trace_moco.py tests/test_wgrad_gpu.py -x -q -m gpu 2>&1 | tail -12) > $O/pytest.log 2>&1
bash tools/probes/pmc_strip.sh > $O/strip_sq_counters.txt 2>&1
(DET=1 timeout 900 python tools/probes/determinism_grads.py 256 32 2>&1 | grep -v "^\[\|Gloo\|^ *$" | tail -20) > $O/det_grads_256.txt
(DET_ONLY=1 timeout 1200 python tools/probes/determinism_check.py 256 32 2>&1 | grep -v "^\[\|Gloo\|^ *$" | tail -8) > $O/det_check_256.txt
for cfg in "3 384" "3 512" "3 768" "2 512" "4 512" "12 256"; do set -- $cfg
  (HCM_BANK_VARIANT=$1 HCM_BANK_ROWS=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $1 rows $2', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])") >> $O/bank_instep.txt
done
(timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --encoder_dtype bf16 2>$O/bf16.err | tail -1) > $O/bench_bf16enc.json
(timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --encoder_dtype bf16 --bank_dtype bf16 --fmap_dtype bf16 --nce_k 131072 2>>$O/bf16.err | tail -1) > $O/bench_config5.json
tail -n 5 $O/pytest.log; tail -4 $O/strip_sq_counters.txt; cat $O/det_grads_256.txt $O/det_check_256.txt $O/bank_instep.txt; for f in bench_bf16enc bench_config5; do python -c "
import json
ls=[l for l in open('$O/$f.json') if l.startswith('{')]
print('$f', (lambda d:(d['value'], d['ms_per_step'], d.get('checked'), d.get('check')))(json.loads(ls[-1])) if ls else open('$O/bf16.err').read()[-1500:])"; done
