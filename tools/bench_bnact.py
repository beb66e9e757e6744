"""GPU time of BatchNorm2d+ReLU (+residual) forward+backward: stock ops vs hcm_bn_act_* (hipEvents)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from hcmoco_amd import hip_ops
dev = 'cuda'
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for shape in [(32, 18, 64, 64), (32, 36, 32, 32), (32, 72, 16, 16), (32, 144, 8, 8), (32, 64, 128, 128), (32, 256, 64, 64), (32, 64, 64, 64)]:
    C = shape[1]
    x = torch.randn(shape, device=dev, requires_grad=True); r = torch.randn(shape, device=dev, requires_grad=True)
    w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev); gy = torch.randn(shape, device=dev)
    def stock():
        y = F.relu(F.batch_norm(x, rm, rv, w, b, True, 0.01, 1e-5) + r, inplace=True); y.backward(gy)
    def fused():
        y = hip_ops.bn_act(x, w, b, rm, rv, 0.01, 1e-5, residual=r, relu=True); y.backward(gy)
    mb = x.numel() * 4 / 1e6
    ts, tf = timeit(stock), timeit(fused)
    print(f'{shape}: {mb:.1f} MB  stock {ts:.1f} us  fused {tf:.1f} us  (fused = {11 * mb / tf * 1e-3:.2f} TB/s over 11 passes)')
