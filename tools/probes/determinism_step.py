"""Which parameter tensors differ bit-wise between two identical runs of N training steps in ONE process
(same seed, same data, fresh model/optimizer/banks each time)?  Run on the GPU box:
    python tools/probes/determinism_step.py [steps]            (HCM_DETERMINISTIC=1 sends every dW to the fixed-order kernels)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hcmoco_amd import _lib
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')


def run():
    args = bench.make_args(int(os.environ.get('PROBE_B', 8)), int(os.environ.get('PROBE_K', 1024)),
                           int(os.environ.get('PROBE_N', 4096)), int(os.environ.get('PROBE_SIZE', 128)), 'coco17', 'nccl',
                           tempfile.mkdtemp(), steps + 1)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    tr = ContrastTrainer(args)
    tr.device = dev
    model, contrast, opt, data = bench.build(args, tr, dev)
    torch.cuda.manual_seed(99)
    it = iter(data)
    losses = [float(tr.train_step(next(it), model, contrast, opt, True)['loss']) for _ in range(steps)]
    torch.cuda.synchronize()
    _lib.torch_glue().set_async_wgrad(False)
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()}, [b.clone() for b in contrast.banks()]


a = run()
b = run()
print('losses', a[0], b[0])
diff = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
print('%d of %d parameter tensors differ bit-wise; banks equal: %s' % (len(diff), len(a[1]),
                                                                       all(torch.equal(x, y) for x, y in zip(a[2], b[2]))))
for n in diff[:40]:
    print('  ', n, tuple(a[1][n].shape), float((a[1][n] - b[1][n]).abs().max()))
