"""Where one un-profiled training step spends its time: hipEvents on the main stream at the phase boundaries of
ContrastTrainer._train_step -- model forward | loss forward | loss.backward() returned | join | optimizer -- plus the
host clock at the same points, averaged over steps.  The events are recorded on the main stream, so a phase's GPU
span ends when the main stream's last kernel of that phase does (the encoders' side work is joined into it).
Usage (GPU box): python tools/probes/phase_times.py"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                             # noqa: E402


def main():
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    if os.environ.get('HCM_FORCE_COLLECTIVES', '0') != '0':        # the N>1 control path on a 1-rank RCCL group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        if os.environ.get('HCM_PROBE_NO_ALLREDUCE'):              # how much of the control path's cost is RCCL itself
            class _Done(object):
                def wait(self):
                    return True
            dist.all_reduce = lambda t, op=None, async_op=False, group=None: _Done()
    args = bench.make_args(32, 16384, 131072, 256, os.environ.get('SKELETON', 'coco17'), 'nccl',
                           tempfile.mkdtemp(), 10 ** 6, arch=os.environ.get('ARCH', 'HRNet'),
                           width=int(os.environ.get('WIDTH', '18')))
    args.rank, args.world_size, args.local_rank, args.gpu = 0, 1, 0, 0
    args.channels_last = False
    trainer = ContrastTrainer(args)
    trainer.device = dev
    model, contrast, opt, data = bench.build(args, trainer, dev)
    it = iter(data)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e, time.perf_counter()))

    net = model.module if hasattr(model, 'module') else model
    fwd = net.forward
    net.forward = lambda *a, **k: (lambda out: (mark('model_forward'), out)[1])(fwd(*a, **k))
    eng = trainer.engine
    for nm in ('bank', 'fmap_sampled', 'fmap', 'section'):
        if hasattr(eng, nm):
            f = getattr(eng, nm)
            setattr(eng, nm, (lambda f, nm: lambda *a, **k: (lambda out: (mark('loss_' + nm), out)[1])(f(*a, **k)))(f, nm))
    # the fused loss section's backward (heads, projection, the eight branch gradients) ends where the encoders'
    # reverse loops begin: model_forward -> loss_section -> section_backward is the serial section of SURVEY 8f-2
    from hcmoco_amd import hip_ops
    bwd0 = hip_ops._Stage2Section.backward
    hip_ops._Stage2Section.backward = staticmethod(lambda ctx, *g: (lambda out: (mark('section_backward'), out)[1])(bwd0(ctx, *g)))
    if trainer.grad_sync is not None:
        red0 = trainer.grad_sync.reduce
        trainer.grad_sync.reduce = lambda join=None: (mark('backward_returned'), red0(join), mark('reduced'))[1]
    join0 = trainer.async_wgrad.wgrad_join if trainer.async_wgrad is not None else None
    if join0 is not None:
        def join():
            if trainer.grad_sync is None:
                mark('backward_returned')
            join0()
            mark('joined')
        trainer.async_wgrad.wgrad_join = join
    step0 = opt.step
    opt.step = lambda *a, **k: (lambda out: (mark('optimizer'), out)[1])(step0(*a, **k))

    for _ in range(10):
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
    torch.cuda.synchronize()
    acc, n = {}, 0
    for _ in range(30):
        marks.clear()
        mark('start')
        trainer.train_step(next(it), model, contrast, opt, stage2=True)
        mark('end')
        torch.cuda.synchronize()
        n += 1
        for (n0, e0, t0), (n1, e1, t1) in zip(marks[:-1], marks[1:]):
            g, h = acc.setdefault(n0 + ' -> ' + n1, [0.0, 0.0])
            acc[n0 + ' -> ' + n1] = [g + e0.elapsed_time(e1), h + (t1 - t0) * 1e3]
    print('%-44s %10s %10s' % ('phase (main-stream events)', 'GPU ms', 'host ms'))
    for k, (g, h) in acc.items():
        print('%-44s %10.2f %10.2f' % (k, g / n, h / n))
    print('sum GPU %.2f ms' % (sum(g for g, _ in acc.values()) / n))


if __name__ == '__main__':
    main()
