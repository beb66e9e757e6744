"""r06 probe: hcm_ball_project_* WITH point features at the shapes of SA levels 2-4 (B = 32): us per call between hipEvents and
the algorithmic bytes (forward: y written once; idx, D, P read twice.  backward: dy, y read twice, dz written; idx, D, P twice)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd import pointnet2_hip as H

d = torch.device('cuda:0')
B = 32
for C1, N, npnt, ns in ((64, 4096, 1024, 16), (64, 4096, 1024, 32), (128, 1024, 256, 16), (128, 1024, 256, 32), (256, 256, 64, 16),
                        (256, 256, 64, 32)):
    torch.manual_seed(ns)
    D = (torch.rand(B, 3, npnt, ns, device=d) - 0.5) * 0.3
    P = torch.randn(B, C1, N, device=d).requires_grad_(True)
    W = torch.randn(C1, 3, device=d).requires_grad_(True)
    gamma = (torch.rand(C1, device=d) + 0.5).requires_grad_(True)
    beta = torch.randn(C1, device=d).requires_grad_(True)
    idx = torch.randint(0, N, (B, npnt, ns), dtype=torch.int32, device=d)
    dy = torch.randn(B, C1, npnt, ns, device=d)

    def fwd():
        return H.ball_project(P, D, W, idx, gamma, beta, None, None, 0.1, 1e-5, True)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    tf = timeit(fwd)
    y = fwd()

    def bwd():
        torch.autograd.grad(y, (P, W, gamma, beta), dy, retain_graph=True)
    tb = timeit(bwd)
    M = B * C1 * npnt * ns
    uniq = 4 * B * npnt * ns + B * C1 * N
    fb, bb = 4 * (M + 2 * uniq), 4 * (5 * M + 2 * uniq)
    print('C1 %3d N %4d np %4d ns %2d: forward %7.1f us (%.2f TB/s of %4d MB)   backward incl. scatter %7.1f us (%.2f TB/s of %4d MB)' % (
        C1, N, npnt, ns, tf, fb / tf / 1e6, fb >> 20, tb, bb / tb / 1e6, bb >> 20))
