set -x
O=gpurun_out/r3_14; mkdir -p $O
(timeout 1200 python -m pytest tests/test_pointnet2_gpu.py -x -q -m gpu 2>&1 | tail -5) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for mode in random nn hub; do python tools/probes/scatter_case.py $mode 2>/dev/null | grep " ms"; done
(timeout 900 python tools/bench_pointnet2.py 2>&1 | grep -v amdgpu.ids) > $O/pointnet2_ops.txt
grep -E "planned|plan |sum of" $O/pointnet2_ops.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; MIOPEN_FIND_MODE=FAST rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no_cpu_baseline --no_check --arch HRNetPN > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python - $(find /tmp/kt -name '*kernel_stats.csv') <<'PY' > $O/hrnetpn_point_ops.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows:
    if any(k in r['Name'] for k in ('scatter', 'plan_kernel', 'three_', 'ball_', 'fps', 'group_', 'gather_points', 'rowmax', 'furthest')):
        print('%-110s %6s calls %9.3f ms  avg %9.1f us' % (r['Name'][:110], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
cat $O/hrnetpn_point_ops.txt
(timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --arch HRNetPN 2>/dev/null | tail -1) > $O/bench_pn18.json
(HCM_PN2_BACKWARD=lds timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_check --arch HRNetPN 2>/dev/null | tail -1) > $O/bench_pn18_lds.json
for f in bench_pn18 bench_pn18_lds; do python -c "
import json
ls=[l for l in open('$O/$f.json') if l.startswith('{')]
print('$f', (lambda d:(d['value'], d['ms_per_step']))(json.loads(ls[-1])) if ls else 'no line')"; done
