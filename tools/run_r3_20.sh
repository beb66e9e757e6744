O=gpurun_out/r3_20; mkdir -p $O
(HCM_BANK_STAGGER=256 timeout 900 python -m pytest tests/test_bank_gpu.py tests/test_memory_module_gpu.py -x -q -m gpu 2>&1 | tail -3)
for rep in 1 2; do for st in 0 128 256 384; do echo -n "stagger $st: "; HCM_BANK_STAGGER=$st python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done
