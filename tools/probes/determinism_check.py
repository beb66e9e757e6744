"""With set_deterministic(True) (every convolution weight gradient on the own fixed-order kernels) is a training
step bit-reproducible, and do the runtimes agree?  Runs the stage-2 trainer for `steps` steps from the same seed
(a) twice with the default runtime, (b) once with plain autograd, (c) once under a 1-rank nccl group, and prints the
largest per-parameter differences.  Decides how far tests/test_trainer_gpu.py and tests/test_rccl_gpu.py can be
tightened.  Usage (GPU box): python tools/probes/determinism_check.py"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                             # noqa: E402


SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def run(plain, steps=3, size=SIZE, batch=BATCH):
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.networks import hrnet
    dev = torch.device('cuda:0')
    args = bench.make_args(batch, 1024, 4096, size, 'coco17', 'nccl', tempfile.mkdtemp(), steps + 1)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    os.environ['HCM_ASYNC_WGRAD'] = '0' if plain else '1'
    hrnet.ENCODER_PROGRAM = not plain
    try:
        tr = ContrastTrainer(args)
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        if plain:
            tr.unwrap(model).two_streams = 0
        torch.cuda.manual_seed(77)
        it = iter(data)
        losses = []
        for _ in range(steps):
            losses.append(float(tr.train_step(next(it), model, contrast, opt, True)['loss']))
        torch.cuda.synchronize()
        params = {n: p.detach().clone() for n, p in tr.unwrap(model).named_parameters()}
        banks = [b.clone() for b in contrast.banks()]
    finally:
        hrnet.ENCODER_PROGRAM = True
        _lib.torch_glue().set_async_wgrad(False)
        os.environ.pop('HCM_ASYNC_WGRAD', None)
    return losses, params, banks


def diff(a, b, tag):
    la, pa, ba = a
    lb, pb, bb = b
    worst, wname, nbit = 0.0, None, 0
    for n in pa:
        if torch.equal(pa[n], pb[n]):
            nbit += 1
            continue
        d = float((pa[n] - pb[n]).abs().max() / pb[n].abs().max().clamp_min(1e-30))
        if d > worst:
            worst, wname = d, n
    print('%-34s losses %s vs %s | %d / %d parameters bit-identical, worst max-rel diff %.3e (%s) | banks equal %s'
          % (tag, ['%.6f' % v for v in la], ['%.6f' % v for v in lb], nbit, len(pa), worst, wname,
             all(torch.equal(x, y) for x, y in zip(ba, bb))), flush=True)


if __name__ == '__main__':
    from hcmoco_amd import _lib
    glue = _lib.torch_glue()
    for det in ((True,) if os.environ.get('DET_ONLY') else (True, False)):
        glue.set_deterministic(det)
        print('=== deterministic weight gradients: %s' % det, flush=True)
        a = run(False)
        b = run(False)
        diff(a, b, 'default runtime, run 1 vs run 2')
        c = run(True)
        diff(a, c, 'default runtime vs plain autograd')
        c2 = run(True)
        diff(c, c2, 'plain autograd, run 1 vs run 2')
