python -c "
from hcmoco_amd.pycontrast.learning import affinity
print('gpu node', affinity.gpu_node(0)[0], len(affinity.gpu_node(0)[1]))"
for rep in 1 2; do
for cfg in "none" "HCM_PIN_NUMA=node" "HCM_PIN_NUMA_FORCE_NODE=0" "HCM_PIN_NUMA_FORCE_NODE=1" "HCM_PIN_NUMA=share"; do
  echo -n "$cfg: "; if [ "$cfg" = none ]; then python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1 > /tmp/l.json; else env $cfg python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_check 2>/dev/null | tail -1 > /tmp/l.json; fi
  python -c "
import json
d=json.loads(open('/tmp/l.json').read()); print(d['value'], d['ms_per_step'], d['config'].get('cpu_affinity'))"
done; done
