"""First layer of an HRNet forward (module path) whose output differs bit-wise between two passes over the
same input with the same weights."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd.pycontrast.networks import hrnet

dev = torch.device('cuda:0')
hrnet.ENCODER_PROGRAM = False
torch.manual_seed(0)
net = hrnet.HighResolutionNet(18).to(dev).train()
x = torch.randn(8, 3, 128, 128, device=dev)
recs = []
for run in range(3):
    rec = []
    hooks = []
    for name, m in net.named_modules():
        if isinstance(m, (hrnet.Conv2d, hrnet.BatchNorm2d, hrnet.ConvBn, hrnet.BasicBlock, hrnet.Bottleneck)):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: rec.append((name, type(mod).__name__, out.detach().clone()))))
    with torch.no_grad():
        ys = net(x)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    rec.append(('OUT0', 'out', ys[0].clone()))
    recs.append(rec)
for other in (1, 2):
    first = None
    ndiff = 0
    for (n0, t0, a), (n1, t1, b) in zip(recs[0], recs[other]):
        if not torch.equal(a, b):
            ndiff += 1
            if first is None:
                first = (n0, t0, tuple(a.shape), float((a - b).abs().max()), float(a.abs().max()))
    print('pass 0 vs pass %d: %d of %d recorded outputs differ; first:' % (other, ndiff, len(recs[0])), first)
# the same with the hcmoco nodes off (stock ATen conv + stock BN) for comparison
hrnet.CONV_GLUE = False
hrnet.FUSED_BN = False
outs = []
with torch.no_grad():
    for _ in range(2):
        outs.append([y.clone() for y in net(x)])
print('stock ATen conv + BN, two passes equal:', [torch.equal(a, b) for a, b in zip(*outs)])
