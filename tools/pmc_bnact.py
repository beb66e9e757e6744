"""Stand-alone launches of the fused normalisation (forward + backward) for rocprofv3 --pmc passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_bnact.py
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_bnact.py
SHAPE=N,C,H,W selects the map (default 32,256,64,64 = 134 MB; 32,18,64,64 = the 9.4 MB branch map)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcmoco_amd import hip_ops

shape = tuple(int(v) for v in os.environ.get('SHAPE', '32,256,64,64').split(','))
d = torch.device('cuda:0')
torch.manual_seed(0)
C = shape[1]
x = torch.randn(shape, device=d, requires_grad=True)
r = torch.randn(shape, device=d, requires_grad=True)
w = torch.ones(C, device=d, requires_grad=True); b = torch.zeros(C, device=d, requires_grad=True)
rm, rv = torch.zeros(C, device=d), torch.ones(C, device=d)
gy = torch.randn(shape, device=d)
for i in range(6):
    y = hip_ops.bn_act(x, w, b, rm, rv, 0.01, 1e-5, residual=r, relu=True)
    torch.autograd.grad(y, (x, r, w, b), gy)
torch.cuda.synchronize()
print('done')
