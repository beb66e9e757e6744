"""Which layers are not run-to-run reproducible?  ONE training step is run twice from the SAME state (weights,
batch-norm buffers, banks, optimizer momentum, sampler offsets restored in between) with set_deterministic(True), and
the parameter gradients of the two runs are compared bit for bit, listed in registration order (encoder1 stem ->
stages -> encoder2 -> SemGCN -> heads).  The gradient of a layer depends on everything downstream of it in the
backward walk, so the LAST layers in forward order that differ name the nondeterministic operation.
Usage (GPU box): python tools/probes/determinism_grads.py [size] [batch]"""
import copy
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                             # noqa: E402


def main():
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    glue = _lib.torch_glue()
    glue.set_deterministic(os.environ.get('DET', '1') != '0')
    dev = torch.device('cuda:0')
    args = bench.make_args(batch, 1024, 4096, size, 'coco17', 'nccl', tempfile.mkdtemp(), 8)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    tr = ContrastTrainer(args)
    tr.device = dev
    model, contrast, opt, data = bench.build(args, tr, dev)
    net = tr.unwrap(model)
    it = iter(data)
    for _ in range(2):                                   # quiet-Find step + one default step: plans and flat buffers exist
        tr.train_step(next(it), model, contrast, opt, True)
    torch.cuda.synchronize()
    batch_ = next(it)
    state = {'model': copy.deepcopy(net.state_dict()), 'banks': [b.clone() for b in contrast.banks()],
             'opt': copy.deepcopy(opt.state_dict()), 'off': contrast.multinomial.offset, 'pix': contrast._pixel_draws}

    def restore():
        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(state['model'][k])
            for b, b0 in zip(contrast.banks(), state['banks']):
                b.copy_(b0)
        opt.load_state_dict(copy.deepcopy(state['opt']))
        contrast.multinomial.offset, contrast._pixel_draws = state['off'], state['pix']

    runs = []
    for _ in range(2):
        restore()
        out = tr.train_step(batch_, model, contrast, opt, True)
        torch.cuda.synchronize()
        runs.append((float(out['loss']), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None},
                     {k: v.detach().clone() for k, v in net.state_dict().items()}))
    glue.set_async_wgrad(False)
    (la, ga, sa), (lb, gb, sb) = runs
    print('losses %.9g %.9g equal %s' % (la, lb, la == lb))
    names = list(ga)
    bad = [n for n in names if not torch.equal(ga[n], gb[n])]
    print('%d / %d parameter gradients differ between two runs of the same step' % (len(bad), len(names)))
    for n in bad[:12] + (['...'] if len(bad) > 24 else []) + bad[-12:]:
        if n == '...':
            print('   ...')
            continue
        d = float((ga[n] - gb[n]).abs().max() / gb[n].abs().max().clamp_min(1e-30))
        print('   %-60s %-18s max-rel %.2e' % (n, tuple(ga[n].shape), d))
    good_after = [n for n in names if n not in set(bad)]
    print('bit-identical gradients: %d, e.g. %s' % (len(good_after), good_after[-6:]))
    badbuf = [k for k in sa if not torch.equal(sa[k], sb[k])]
    print('%d / %d state_dict entries differ after the step, first: %s' % (len(badbuf), len(sa), badbuf[:6]))


if __name__ == '__main__':
    main()
