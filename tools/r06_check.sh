O=gpurun_out/r6_a; mkdir -p $O
(timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -40) > $O/pytest_full.log 2>&1
tail -3 $O/pytest_full.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3)
python bench.py 2>/dev/null | grep "^{" | tail -1 > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d.get('checked'), d['roofline']['frac']); 
for k,v in d.get('roofline_secondary',{}).items(): print(k, v.get('avg_launch_ms'), v.get('frac'))
"
