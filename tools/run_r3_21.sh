O=gpurun_out/r3_21; mkdir -p $O
(timeout 1200 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no_cpu_baseline > $O/two_ranks.log 2>&1; echo rc=$?)
tail -1 $O/two_ranks.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['checked'], d['config']['grad_collectives_per_step'], d['config']['backend'])" || tail -20 $O/two_ranks.log
(HCM_FORCE_COLLECTIVES=1 timeout 900 python bench.py --steps 5 --warmup 2 --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('forced', d['value'], d['checked'], d['config']['grad_collectives_per_step'], d['config']['backend'])")
for rep in 1 2; do for rows in 512 1024; do echo -n "rows $rows: "; HCM_BANK_ROWS=$rows python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done
