"""Does MIOpen run the two HRNets' convolutions as one groups=2 convolution at a sane speed?
Prints host enqueue time per layer; run under rocprofv3 --kernel-trace --stats for the GPU side."""
import os, sys, time
import torch, torch.nn as nn
dev = 'cuda'
which, C, hw, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
N = 40
if which == 'pair':
    convs = [nn.Conv2d(C, C, k, 1, k // 2, bias=False).to(dev) for _ in range(2)]
    xs = [torch.randn(32, C, hw, hw, device=dev, requires_grad=True) for _ in range(2)]
    def run():
        ys = list(xs)
        for _ in range(N):
            ys = [c(y) for c, y in zip(convs, ys)]
        return ys[0].sum() + ys[1].sum()
else:
    conv = nn.Conv2d(2 * C, 2 * C, k, 1, k // 2, groups=2, bias=False).to(dev)
    x = torch.randn(32, 2 * C, hw, hw, device=dev, requires_grad=True)
    def run():
        y = x
        for _ in range(N):
            y = conv(y)
        return y.sum()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l = run(); t1 = time.perf_counter(); l.backward(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
print(f'{which} C={C} hw={hw} k={k}: host fwd {1e6*(t1-t0)/N:.1f} bwd {1e6*(t2-t1)/N:.1f} us/layer(pair)  wall {1e3*(t3-t0)/N*1e3:.1f} us/layer(pair)')
