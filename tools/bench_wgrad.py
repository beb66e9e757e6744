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

# 1x1 convolutions of the fuse layers / bottlenecks: (N, C_in, K_out, H)
for N, C, K, H in [(32, 36, 18, 32), (32, 72, 18, 16), (32, 144, 18, 8), (32, 72, 36, 16), (32, 144, 36, 8), (32, 144, 72, 8),
                   (32, 64, 256, 64), (32, 256, 64, 64)]:
    W = H
    x = torch.randn(N, C, H, W, device=dev); dy = torch.randn(N, K, H, W, device=dev); w = torch.randn(K, C, 1, 1, device=dev)
    args = (dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
    ref = torch.ops.aten.convolution_backward(*args)[1]
    nb = int(L.hcm_conv1x1_wgrad_workspace_bytes(N, C, K, H, W))
    if nb == 0:
        print(f'1x1 N={N} C={C} K={K} {H}x{W}: outside the kernel\'s geometry'); continue
    got = hip_ops.conv3x3_wgrad(x, dy, ksize=1)
    err = float((got - ref).abs().max() / ref.abs().max())
    ws = torch.empty(nb, dtype=torch.uint8, device=dev); dw = torch.empty_like(w)
    st = CT.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(*args))
    t_new = timeit(lambda: L.hcm_conv1x1_wgrad(x.data_ptr(), dy.data_ptr(), N, C, K, H, W, dw.data_ptr(), ws.data_ptr(), nb, st))
    print(f'1x1 N={N} C={C} K={K} {H}x{W}: rel err {err:.2e}  MIOpen {t_ref:.1f} us  hcm {t_new:.1f} us')

# 3x3 stride-2 convolutions of the fuse layers / transitions: (N, C_in, K_out, H_in)
for N, C, K, H in [(32, 18, 18, 64), (32, 18, 36, 64), (32, 36, 36, 32), (32, 36, 72, 32), (32, 72, 144, 16), (32, 18, 72, 32)]:
    W = H
    x = torch.randn(N, C, H, W, device=dev); dy = torch.randn(N, K, H // 2, W // 2, device=dev); w = torch.randn(K, C, 3, 3, device=dev)
    args = (dy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
    ref = torch.ops.aten.convolution_backward(*args)[1]
    nb = int(L.hcm_conv3x3s2_wgrad_workspace_bytes(N, C, K, H // 2, W // 2))
    if nb == 0:
        print(f's2 N={N} C={C} K={K} {H}x{W}: outside the kernel\'s geometry'); continue
    got = hip_ops.conv3x3_wgrad(x, dy, ksize=3, stride=2)
    err = float((got - ref).abs().max() / ref.abs().max())
    ws = torch.empty(nb, dtype=torch.uint8, device=dev); dw = torch.empty_like(w)
    st = CT.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(*args))
    t_new = timeit(lambda: L.hcm_conv3x3s2_wgrad(x.data_ptr(), dy.data_ptr(), N, C, K, H // 2, W // 2, dw.data_ptr(), ws.data_ptr(), nb, st))
    print(f's2 N={N} C={C} K={K} {H}x{W}: rel err {err:.2e}  MIOpen {t_ref:.1f} us  hcm {t_new:.1f} us')
