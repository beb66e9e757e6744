"""Host-side (enqueue) time of the phases of one stage-2 step: the GPU runs behind the host, so the
wall time of each phase WITHOUT synchronisation is what the Python/launch path costs."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer

dev = torch.device('cuda:0')
args = bench.make_args(32, 16384, 131072, 256, 'coco17', 'nccl', tempfile.mkdtemp(), 100)
args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
tr = ContrastTrainer(args); tr.device = dev
model, contrast, opt, data = bench.build(args, tr, dev)
it = iter(data)
for _ in range(5):
    tr.train_step(next(it), model, contrast, opt, True)
torch.cuda.synchronize()
T = dict(fwd=0.0, gather=0.0, bank=0.0, fmap=0.0, bwd=0.0, join=0.0, opt=0.0)
N = 15
for _ in range(N):
    d = next(it)
    t0 = time.perf_counter()
    f1, f2, f3, f, aux = model(d[0], d[2], return_fm=True)
    t1 = time.perf_counter()
    all_f, all_i = tr._packed_gather(f, d[1])
    a, b, c = torch.chunk(f, 3, dim=1); A, B_, C_ = torch.chunk(all_f, 3, dim=1)
    t2 = time.perf_counter()
    total, losses, accs = tr.engine.bank(contrast, a, b, c, d[1], A, B_, C_, all_i, use_depth=d[6])
    t3 = time.perf_counter()
    net = tr.unwrap(model)
    ft, m = tr.engine.fmap_sampled(f1, f2, net.encoder1_linear, net.encoder2_linear, f3, d[7], d[4], d[5], d[6], None, 400, 0.07)
    t4 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    (total + ft).backward()
    t5 = time.perf_counter()
    if tr.async_wgrad is not None:
        tr.async_wgrad.wgrad_join()
    t5b = time.perf_counter()
    opt.step()
    t6 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5b - t5, t6 - t5b)):
        T[k] += v
    torch.cuda.synchronize()
print('host enqueue ms per step:', {k: round(1e3 * v / N, 2) for k, v in T.items()}, 'sum', round(1e3 * sum(T.values()) / N, 2))
