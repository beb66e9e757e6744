"""csrc/conv1x1.hip (forward, data gradient, weight gradient of the SharedMLP 1x1 convolutions) against ATen / MIOpen on the ball
tensors of an HRNetPN step (B = 32; pointnet2_msg.py NPOINTS / NSAMPLE / MLPS): us per call between hipEvents (20 calls),
the fp32 MFMA floor (2 N P C K flops at 157.3 TF) and the HBM floor (operands once at 8 TB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as CT
import torch
from hcmoco_amd import _lib
L = _lib.lib()
if 'ARITH' in os.environ:                     # 0 = default (split-bf16 from 64 channels up), 1 = exact fp32 everywhere
    L.hcm_conv1x1_set_arith(int(os.environ['ARITH']))
dev = 'cuda'


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(True); b = torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


p = lambda t: CT.c_void_p(t.data_ptr())
B = int(os.environ.get('B', 32))
SHAPES = [(16, 32, 4096, 16), (32, 64, 4096, 32), (64, 128, 1024, 16), (64, 128, 1024, 32), (128, 256, 256, 32)]
if 'SHAPES' in os.environ:                    # e.g. SHAPES=32,64,4104,32;64,128,1032,32 (camping probe: row strides off the powers of two)
    SHAPES = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
if 'ONLY' in os.environ:                      # tools/probes/pmc_conv1x1.sh: one layer under the counters
    SHAPES = [SHAPES[int(os.environ['ONLY'])]]
for C, K, npnt, ns in SHAPES:
    P = npnt * ns
    x = torch.randn(B, C, npnt, ns, device=dev); dy = torch.randn(B, K, npnt, ns, device=dev); w = torch.randn(K, C, 1, 1, device=dev)
    z, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
    st = CT.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = int(L.hcm_conv1x1_ball_wgrad_workspace_bytes(B, C, K, npnt, ns)); ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
    bw = lambda m: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, m)
    t = dict(fwd_hcm=timeit(lambda: L.hcm_conv1x1_forward(p(x), p(w), p(z), B, C, K, P, st)),
             dx_hcm=timeit(lambda: L.hcm_conv1x1_backward_data(p(dy), p(w), p(dx), B, C, K, P, st)),
             dw_hcm=timeit(lambda: L.hcm_conv1x1_ball_wgrad(p(x), p(dy), B, C, K, npnt, ns, p(dw), p(ws), nb, st)))
    if os.environ.get('ATEN', '0') == '1':            # MIOpen's Find on these shapes takes minutes
        t.update(fwd_aten=timeit(lambda: torch.nn.functional.conv2d(x, w)), dx_aten=timeit(lambda: bw([True, False, False])),
                 dw_aten=timeit(lambda: bw([False, True, False])))
    ref = torch.einsum('nkp,ncp->kc', dy.reshape(B, K, -1), x.reshape(B, C, -1)).reshape(K, C, 1, 1)
    err = float((dw - ref).abs().max() / ref.abs().max())
    mfma = 2.0 * B * P * C * K / 157.3e12 * 1e6
    hbm = 4.0 * B * P * (C + K) / 8e12 * 1e6
    print(f'{C:>3} -> {K:<3} P={P:<6} ' + '  '.join(f'{k} {v:7.1f}' for k, v in t.items()) +
          f'  | floors: mfma {mfma:.0f} us, hbm {hbm:.0f} us | dw vs einsum {err:.1e}', flush=True)
