"""When the HRNet and the cloud branch of the HRNetPN model run on the GPU relative to each other (un-profiled, HIP events
on their own streams; ``net.trace_streams = True`` in networks/build_backbone.py).  usage: hrnetpn_streams.py"""
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer

dev = torch.device('cuda:0')
args = bench.make_args(32, 16384, 131072, 256, 'coco17', 'nccl', tempfile.mkdtemp(), 10 ** 6, arch='HRNetPN', width=18)
args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
tr = ContrastTrainer(args)
tr.device = dev
model, contrast, opt, data = bench.build(args, tr, dev)
net = tr.unwrap(model)
net.trace_streams = True
it = iter(data)
for _ in range(8):
    tr.train_step(next(it), model, contrast, opt, stage2=True)
torch.cuda.synchronize()
acc = {}
n = 0
for _ in range(20):
    t0 = time.perf_counter()
    tr.train_step(next(it), model, contrast, opt, stage2=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    ev = net._stream_trace
    row = {'step_ms': dt, 'hrnet_start': ev['t0'].elapsed_time(ev['h0']), 'hrnet_end': ev['t0'].elapsed_time(ev['h1']),
           'cloud_start': ev['t0'].elapsed_time(ev['p0']), 'cloud_end': ev['t0'].elapsed_time(ev['p1'])}
    for k, v in row.items():
        acc[k] = acc.get(k, 0.0) + v
    n += 1
print('two_streams = %d (ms after the forward began on the GPU; synchronised steps):' % net.two_streams,
      {k: round(v / n, 2) for k, v in acc.items()})
