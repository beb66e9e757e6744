"""r06 probe: hcm_ball_project_* at the FIRST SA level's shapes (no point features; B = 32, 4096 centres, 16 / 32 members, 16 / 32
channels) -- us per call between hipEvents and the algorithmic bytes (forward: y written once + D read twice; backward: dy, y
read once + D)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd import pointnet2_hip as H

d = torch.device('cuda:0')
B, npnt = 32, 4096
for C1, ns in ((16, 16), (32, 32)):
    torch.manual_seed(ns)
    D = (torch.rand(B, 3, npnt, ns, device=d) - 0.5) * 0.1
    W = torch.randn(C1, 3, device=d).requires_grad_(True)
    gamma = (torch.rand(C1, device=d) + 0.5).requires_grad_(True)
    beta = torch.randn(C1, device=d).requires_grad_(True)
    idx = torch.zeros(B, npnt, ns, dtype=torch.int32, device=d)
    dy = torch.randn(B, C1, npnt, ns, device=d)

    def fwd():
        return H.ball_project(None, D, W, idx, gamma, beta, None, None, 0.1, 1e-5, True)

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
        torch.autograd.grad(y, (W, gamma, beta), dy, retain_graph=True)
    tb = timeit(bwd)
    M = B * C1 * npnt * ns
    fb, bb = 4 * (M + 2 * 3 * B * npnt * ns), 4 * (2 * M + 3 * B * npnt * ns)
    print('C1 %2d ns %2d: forward %7.1f us (%.2f TB/s of %d MB)   backward %7.1f us (%.2f TB/s of %d MB)' % (
        C1, ns, tf, fb / tf / 1e6, fb >> 20, tb, bb / tb / 1e6, bb >> 20))
