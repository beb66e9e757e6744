O=gpurun_out/r3_17; mkdir -p $O
(timeout 1200 python -m pytest tests/test_pointnet2_gpu.py -x -q -m gpu 2>&1 | tail -5) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for mode in random nn hub; do python tools/probes/scatter_case.py $mode 2>/dev/null | grep " ms"; done
(timeout 900 python tools/bench_pointnet2.py 2>&1 | grep -v amdgpu.ids) > $O/pointnet2_ops.txt
grep -E "planned|sum of" $O/pointnet2_ops.txt
(timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --arch HRNetPN 2>/dev/null | tail -1) > $O/bench_pn18.json
python -c "
import json
ls=[l for l in open('$O/bench_pn18.json') if l.startswith('{')]
print('bench_pn18', (lambda d:(d['value'], d['ms_per_step']))(json.loads(ls[-1])) if ls else 'no line')"
