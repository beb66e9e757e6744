import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hcmoco_amd.pycontrast.networks.hrnet import HighResolutionNet
from hcmoco_amd import _lib
dev = torch.device('cuda:0')
net = HighResolutionNet(18).to(dev).train()
x = torch.randn(2, 3, 64, 64, device=dev)
maps = net(x)
sum(m.square().mean() for m in maps).backward()
_lib.torch_glue().wgrad_join()
ps = list(net.last_program.params)
print('params', len(ps), 'ids', len(set(id(p) for p in ps)), 'model params', len(list(net.parameters())))
g = ps[0].grad
print('grad none', sum(1 for p in ps if p.grad is None))
print('base', None if g._base is None else (g._base.shape, g._base.storage_offset()), 'g off', g.storage_offset(), 'contig', g.is_contiguous())
n = sum(p.numel() for p in ps)
print('n', n, 'base numel', None if g._base is None else g._base.numel())
o = 0
bad = 0
for i, p in enumerate(ps):
    gg = p.grad
    if gg._base is not g._base or gg.storage_offset() != o:
        if bad < 5:
            print('mismatch at', i, tuple(p.shape), gg.storage_offset(), 'expected', o, gg._base is g._base)
        bad += 1
    o += p.numel()
print('bad', bad)
