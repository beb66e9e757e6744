# What an FPS round is bound by (VERDICT r03 #3d): SQ counters of fps_kernel at n = m = 4096, B = 32 (one workgroup of 512
# threads per cloud, 4095 serial rounds).  usage (GPU box): bash tools/probes/pmc_fps.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fps_only.py <<PY
import sys
sys.path.insert(0, '$R')
import torch
from hcmoco_amd import pointnet2_hip as P
d = torch.device('cuda:0')
torch.manual_seed(0)
xyz = torch.rand(32, 4096, 3, device=d)
for _ in range(3):
    out = torch.zeros(32, 4096, dtype=torch.int32, device=d)
    temp = torch.full((32, 4096), 1e10, device=d)
    P.furthest_point_sampling_wrapper(32, 4096, 4096, xyz, temp, out)
torch.cuda.synchronize()
PY
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pf
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pf -o p -- python /tmp/fps_only.py > /tmp/pf.log 2>&1 || tail -3 /tmp/pf.log
  python - $(find /tmp/pf -name '*counter_collection.csv') <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if 'fps_kernel' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('fps_kernel n=m=4096 B=32:', {c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
done
