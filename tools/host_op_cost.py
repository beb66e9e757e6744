"""Host enqueue cost per BatchNorm+ReLU layer (forward, backward): stock ops (MIOpen) vs the C++ autograd node
(torch.ops.hcmoco.bn_act); r01 also measured a Python autograd.Function over the same entry points (deleted in r05)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from hcmoco_amd import hip_ops
dev = 'cuda'
x0 = torch.randn(32, 18, 64, 64, device=dev, requires_grad=True)
bn = nn.BatchNorm2d(18).to(dev)
N = 300
args = (bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.1, 1e-5)

def chain(layer):
    x = x0
    for _ in range(N):
        x = layer(x)
    return x

variants = {
    'stock': lambda x: F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, 0.1, 1e-5), inplace=True),
    'c++-node': lambda x: hip_ops._lib.torch_glue().bn_act(x, None, *args, True),
}
for name, layer in variants.items():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = chain(layer); t1 = time.perf_counter()
        y.sum().backward(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'{name}: fwd {1e6*(t1-t0)/N:.1f} us/layer  bwd {1e6*(t2-t1)/N:.1f} us/layer  total wall {1e3*(t3-t0):.1f} ms')

# the convolutions either side of the normalisation (MIOpen through ATen)
for cin, hw, k in ((18, 64, 3), (36, 32, 3), (144, 8, 3), (64, 64, 1)):
    conv = nn.Conv2d(cin, cin, k, 1, k // 2, bias=False).to(dev)
    xc = torch.randn(32, cin, hw, hw, device=dev, requires_grad=True)
    def chain_conv():
        x = xc
        for _ in range(N):
            x = conv(x)
        return x
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = chain_conv(); t1 = time.perf_counter()
        y.sum().backward(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'ATen conv{k}x{k} {cin}ch @{hw}: fwd {1e6*(t1-t0)/N:.1f} us/layer  bwd {1e6*(t2-t1)/N:.1f} us/layer  total wall {1e3*(t3-t0):.1f} ms')

    op = hip_ops._lib.torch_glue().conv2d.default
    w = conv.weight
    def chain_glue():
        x = xc
        for _ in range(N):
            x = op(x, w, 1, k // 2)
        return x
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = chain_glue(); t1 = time.perf_counter()
        y.sum().backward(); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'glue conv{k}x{k} {cin}ch @{hw}: fwd {1e6*(t1-t0)/N:.1f} us/layer  bwd {1e6*(t2-t1)/N:.1f} us/layer  total wall {1e3*(t3-t0):.1f} ms')
