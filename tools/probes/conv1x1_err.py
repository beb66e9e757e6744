"""conv1x1 forward / data gradient against float64 on the shapes of tests/test_pointnet2_gpu.py (max abs error / max |reference|)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as Cc
import torch
from hcmoco_amd import _lib
L = _lib.lib()
if 'ARITH' in os.environ:                     # 0 = default (split-bf16 from 64 channels up), 1 = exact fp32 everywhere
    L.hcm_conv1x1_set_arith(int(os.environ['ARITH']))
dev = torch.device('cuda:0')
p = lambda t: Cc.c_void_p(t.data_ptr())
for N, C, K, H, W in [(2, 16, 32, 128, 16), (3, 32, 64, 64, 32), (2, 64, 128, 32, 16), (2, 256, 512, 16, 16), (2, 128, 256, 24, 8),
                      (32, 32, 64, 4096, 32), (2, 32, 16, 64, 8), (1, 160, 48, 40, 8), (5, 64, 128, 1024, 16)]:
    torch.manual_seed(C * 7 + K)
    P = H * W
    x = torch.randn(N, C, H, W, device=dev); w = torch.randn(K, C, 1, 1, device=dev) / C ** 0.5; g = torch.randn(N, K, H, W, device=dev)
    z, dx = torch.full((N, K, H, W), float('nan'), device=dev), torch.full((N, C, H, W), float('nan'), device=dev)
    st = Cc.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.hcm_conv1x1_forward(p(x), p(w), p(z), N, C, K, P, st) == 0
    assert L.hcm_conv1x1_backward_data(p(g), p(w), p(dx), N, C, K, P, st) == 0
    torch.cuda.synchronize()
    w2 = w.view(K, C).double()
    zr = torch.matmul(w2, x.view(N, C, P).double()).view(N, K, H, W)
    dxr = torch.matmul(w2.t(), g.view(N, K, P).double()).view(N, C, H, W)
    ez = float((z.double() - zr).abs().max() / zr.abs().max()); ed = float((dx.double() - dxr).abs().max() / dxr.abs().max())
    rz = float((z.double() - zr).norm() / zr.norm()); rd = float((dx.double() - dxr).norm() / dxr.norm())
    print(f'{N:>2} {C:>3}->{K:<3} P={P:<6} fwd max {ez:.2e} l2 {rz:.2e}   dx max {ed:.2e} l2 {rd:.2e}', flush=True)
