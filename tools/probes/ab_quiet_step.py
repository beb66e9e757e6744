import os, sys, tempfile, time, warnings
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
quiet = sys.argv[1] == 'quiet'
dev = torch.device('cuda:0')
args = bench.make_args(32, 16384, 131072, 256, 'coco17', 'nccl', tempfile.mkdtemp(), 10 ** 6, arch='HRNetPN', width=18)
args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
tr = ContrastTrainer(args); tr.device = dev
if not quiet:
    tr._find_done = True
model, contrast, opt, data = bench.build(args, tr, dev)
it = iter(data)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    for _ in range(10):
        tr.train_step(next(it), model, contrast, opt, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.train_step(next(it), model, contrast, opt, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(sys.argv[1], 'samples/s %.1f' % (32 * 30 / dt), 'accumulate-grad warnings:', sum('AccumulateGrad' in str(x.message) for x in w))
