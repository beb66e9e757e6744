set -x
O=gpurun_out/r3_12; mkdir -p $O
(timeout 1200 python -m pytest tests/test_pointnet2_gpu.py tests/test_hrnetpn.py -x -q -m gpu 2>&1 | tail -15) > $O/pytest.log 2>&1
(timeout 900 python tools/bench_pointnet2.py 2>&1 | grep -v amdgpu.ids) > $O/pointnet2_ops.txt
(timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --arch HRNetPN 2>/dev/null | tail -1) > $O/bench_pn18.json
(HCM_PN2_BACKWARD=lds timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --arch HRNetPN 2>/dev/null | tail -1) > $O/bench_pn18_lds.json
tail -n 8 $O/pytest.log; cat $O/pointnet2_ops.txt | grep -E "grad|plan|sum of"; for f in bench_pn18 bench_pn18_lds; do python -c "
import json
ls=[l for l in open('$O/$f.json') if l.startswith('{')]
print('$f', (lambda d:(d['value'], d['ms_per_step']))(json.loads(ls[-1])) if ls else 'no line')"; done
