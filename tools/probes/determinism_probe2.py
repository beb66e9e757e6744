"""Bit-wise repeatability of the pieces of one training step on the GPU box:
  (1) torch.ops.hcmoco.conv2d forward / data gradient / weight gradient per HRNet layer shape (MIOpen or own kernels),
  (2) the gradients of ONE stage-2 step, per parameter, between two identical runs."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hcmoco_amd import _lib
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer

dev = torch.device('cuda:0')
ops = _lib.torch_glue()
print('== conv2d node, two calls each ==')
shapes = [(32, 3, 64, 256, 3, 2), (32, 64, 64, 128, 3, 2), (32, 64, 64, 64, 1, 1), (32, 64, 64, 64, 3, 1), (32, 64, 256, 64, 1, 1),
          (32, 256, 64, 64, 1, 1), (32, 18, 18, 64, 3, 1), (32, 36, 36, 32, 3, 1), (32, 72, 72, 16, 3, 1), (32, 144, 144, 8, 3, 1),
          (32, 36, 18, 32, 1, 1), (32, 144, 18, 8, 1, 1), (32, 18, 36, 64, 3, 2), (32, 72, 144, 16, 3, 2), (32, 256, 18, 64, 3, 1)]
for (N, C, K, H, ks, st) in shapes:
    g = torch.Generator(device='cpu').manual_seed(C * K + H)
    x = torch.randn(N, C, H, H, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(K, C, ks, ks, generator=g) * 0.05).to(dev).requires_grad_(True)
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = ops.conv2d(x, w, st, ks // 2)
        gy = torch.sin(torch.arange(y.numel(), device=dev, dtype=torch.float32)).view_as(y)
        y.backward(gy)
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    same = [torch.equal(a, b) for a, b in zip(*outs)]
    print('N%d C%d K%d H%d k%d s%d  fwd %s  dX %s  dW %s' % (N, C, K, H, ks, st, *same))

print('== one stage-2 step, gradients of two identical runs ==')


def run():
    args = bench.make_args(8, 1024, 4096, 128, 'coco17', 'nccl', tempfile.mkdtemp(), 2)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    tr = ContrastTrainer(args)
    tr.device = dev
    model, contrast, opt, data = bench.build(args, tr, dev)
    torch.cuda.manual_seed(99)
    opt.step = lambda *a, **k: None                    # keep the gradients, skip the update
    tr.train_step(next(iter(data)), model, contrast, opt, True)
    torch.cuda.synchronize()
    _lib.torch_glue().set_async_wgrad(False)
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


a, b = run(), run()
names = list(a)
diff = [n for n in names if not torch.equal(a[n], b[n])]
print('%d of %d gradient tensors differ' % (len(diff), len(names)))
top = {}
for n in names:
    top.setdefault(n.split('.')[0], [0, 0])
    top[n.split('.')[0]][0] += 1
    top[n.split('.')[0]][1] += n in diff
print(top)
enc = [n for n in names if n.startswith('encoder1.')]
print('encoder1 tensors in registration order, first/last differing:',
      next((n for n in enc if n in diff), None), '|', next((n for n in reversed(enc) if n in diff), None))
print('encoder1 identical tensors:', [n for n in enc if n not in diff][:12])
