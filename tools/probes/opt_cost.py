"""How much of the step is the optimizer?  Same stage-2 step timed (a) as is, (b) with optimizer.step() skipped,
(c) host time of optimizer.step() alone.  Run on the GPU box."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer

dev = torch.device('cuda:0')
args = bench.make_args(32, 16384, 131072, 256, 'coco17', 'nccl', tempfile.mkdtemp(), 200)
args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
tr = ContrastTrainer(args)
tr.device = dev
model, contrast, opt, data = bench.build(args, tr, dev)
it = iter(data)
for _ in range(6):
    tr.train_step(next(it), model, contrast, opt, True)


def timed(n=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(next(it), model, contrast, opt, True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print('step as is            %.2f ms' % timed())
real = opt.step
host = []


def measured(*a, **k):
    t0 = time.perf_counter()
    r = real(*a, **k)
    host.append(time.perf_counter() - t0)
    return r


opt.step = measured
print('step (timing opt)     %.2f ms' % timed(), ' optimizer.step host time %.2f ms' % (1e3 * sum(host) / len(host)))
zg = opt.zero_grad
zt = []


def zmeasured(*a, **k):
    t0 = time.perf_counter()
    r = zg(*a, **k)
    zt.append(time.perf_counter() - t0)
    return r


opt.zero_grad = zmeasured
opt.step = lambda *a, **k: None
print('step without opt.step %.2f ms' % timed(), ' zero_grad host time %.2f ms' % (1e3 * sum(zt) / len(zt)))
