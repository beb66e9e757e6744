"""Bit-wise repeatability of the library GEMMs / ATen ops around the encoders at the step's shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
if len(sys.argv) > 1 and sys.argv[1] == 'det':
    torch.use_deterministic_algorithms(True, warn_only=True)
    print('use_deterministic_algorithms(True, warn_only=True)')


def twice(name, fn, *leaves):
    outs = []
    for _ in range(2):
        for t in leaves:
            t.grad = None
        y = fn()
        g = torch.sin(torch.arange(y.numel(), device=dev, dtype=torch.float32)).view_as(y)
        y.backward(g)
        torch.cuda.synchronize()
        outs.append([y.detach().clone()] + [t.grad.clone() for t in leaves])
    print('%-46s fwd %s  grads %s' % (name, torch.equal(outs[0][0], outs[1][0]),
                                      [torch.equal(a, b) for a, b in zip(outs[0][1:], outs[1][1:])]))


torch.manual_seed(0)
for B in (8, 32):
    x = torch.randn(B, 270, device=dev, requires_grad=True)
    lin = torch.nn.Linear(270, 128).to(dev)
    twice('Linear(270,128) B=%d' % B, lambda: lin(x), x, lin.weight, lin.bias)
    x3 = torch.randn(B, 128, device=dev, requires_grad=True)
    lin3 = torch.nn.Linear(128, 128).to(dev)
    twice('Linear(128,128) B=%d' % B, lambda: lin3(x3), x3, lin3.weight, lin3.bias)
    rows = torch.randn(B, 417, 270, device=dev, requires_grad=True)
    w = torch.randn(128, 270, device=dev, requires_grad=True)
    bb = torch.randn(128, device=dev, requires_grad=True)
    twice('F.linear rows [B,417,270] B=%d' % B, lambda: F.linear(rows, w, bb), rows, w, bb)
    for hw, c in ((1024, 36), (256, 72), (64, 144)):
        S = torch.rand(B, 417, hw, device=dev)
        m = torch.randn(B, c, hw, device=dev, requires_grad=True)
        twice('bmm(S[B,417,%d], map^T c=%d) B=%d' % (hw, c, B), lambda: torch.bmm(S, m.transpose(1, 2)), m)
    xs = torch.randn(B * 17, 128, device=dev, requires_grad=True)
    wc = torch.randn(128, 256, device=dev, requires_grad=True)
    twice('mm [B*17,128]x[128,256] B=%d' % B, lambda: torch.mm(xs, wc), xs, wc)
    fm = torch.randn(B, 18, 64, 64, device=dev, requires_grad=True)
    twice('mean over (2,3) B=%d' % B, lambda: fm.mean((2, 3)), fm)
    v = torch.randn(B, 128, device=dev, requires_grad=True)
    twice('F.normalize B=%d' % B, lambda: F.normalize(v), v)
