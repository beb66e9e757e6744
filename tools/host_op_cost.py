"""Host enqueue cost of native BN+ReLU (MIOpen) vs a Python autograd.Function issuing two C-ABI launches.
Decides whether a fused BN+ReLU kernel can pay for itself on the host side."""
import time, torch, torch.nn as nn, torch.nn.functional as F
import ctypes as C
from hcmoco_amd import hip_ops, _lib
L = _lib.lib()
def launch(a, b):
    hip_ops.check(L.hcm_upsample_bilinear2d(C.c_void_p(a.data_ptr()), 32 * 18, 64, 64, 64, 64, C.c_void_p(b.data_ptr()), hip_ops._stream()), 'x')
dev = 'cuda'
x0 = torch.randn(32, 18, 64, 64, device=dev, requires_grad=True)
bn = nn.BatchNorm2d(18).to(dev)
N = 300

def native():
    x = x0
    for _ in range(N):
        x = F.relu(bn(x), inplace=True)
    return x

class TwoLaunch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        y = torch.empty_like(x); m = torch.empty(18, device=x.device); v = torch.empty(18, device=x.device)
        launch(x, y); launch(x, y)
        ctx.save_for_backward(x, y, w, m, v)
        return y
    @staticmethod
    def backward(ctx, g):
        x, y, w, m, v = ctx.saved_tensors
        gx = torch.empty_like(g); gw = torch.empty(18, device=g.device); gb = torch.empty(18, device=g.device)
        launch(x, gx); launch(x, gx)
        return gx, gw, gb

def custom():
    x = x0
    for _ in range(N):
        x = TwoLaunch.apply(x, bn.weight, bn.bias)
    return x

for name, fn in (('native', native), ('custom', custom)):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = fn(); t1 = time.perf_counter()
        y.sum().backward(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'{name}: fwd {1e6*(t1-t0)/N:.1f} us/layer  bwd {1e6*(t2-t1)/N:.1f} us/layer  total wall {1e3*(t3-t0):.1f} ms')
