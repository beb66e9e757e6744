"""hcm_conv3x3_wgrad against MIOpen's backward-weights (through ATen) on the HRNet shapes: error and
wall time per call (100 back-to-back calls between hipEvents; MIOpen's figure includes its transposes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as CT
from hcmoco_amd import hip_ops, _lib
L = _lib.lib()
dev = 'cuda'
def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, C, H in [(32, 18, 64), (32, 36, 32), (32, 72, 16), (32, 144, 8), (32, 32, 64), (32, 64, 32), (32, 128, 16), (32, 256, 8), (32, 64, 64), (5, 7, 12)]:
    W = H if H != 12 else 20
    x = torch.randn(N, C, H, W, device=dev); dy = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev)
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    got = hip_ops.conv3x3_wgrad(x, dy)
    err = float((got - ref).abs().max() / ref.abs().max())
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    nb = int(L.hcm_conv3x3_wgrad_workspace_bytes(N, C, C, H, W)); ws = torch.empty(nb, dtype=torch.uint8, device=dev); dw = torch.empty_like(w)
    st = CT.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_new = timeit(lambda: L.hcm_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), N, C, C, H, W, dw.data_ptr(), ws.data_ptr(), nb, st))
    print(f'N={N} C=K={C} {H}x{W}: rel err {err:.2e}  MIOpen {t_ref:.1f} us  hcm {t_new:.1f} us')
