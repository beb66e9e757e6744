"""Sum kernel time per variant from a rocprofv3 kernel trace of tools/bench_bnact.py-like loops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from hcmoco_amd import hip_ops
dev = 'cuda'
which = sys.argv[1]
shape = tuple(int(v) for v in sys.argv[2].split(','))
C = shape[1]
x = torch.randn(shape, device=dev, requires_grad=True); r = torch.randn(shape, device=dev, requires_grad=True)
w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev); gy = torch.randn(shape, device=dev)
for _ in range(20):
    if which == 'stock':
        y = F.relu(F.batch_norm(x, rm, rv, w, b, True, 0.01, 1e-5) + r, inplace=True)
    else:
        y = hip_ops.bn_act(x, w, b, rm, rv, 0.01, 1e-5, residual=r, relu=True)
    gx, gr, gw, gb = torch.autograd.grad(y, (x, r, w, b), gy)
torch.cuda.synchronize()
