"""DDP training for contrastive pre-training -- entry point.

Same flow as the reference's ``main_contrast.py`` (/root/reference/pycontrast/main_contrast.py:
19-106): parse -> trainer + process group -> model -> loader -> memory bank -> (--pretrain) ->
SGD -> wrap -> broadcast banks -> resume -> epoch loop.  Launch one process per GPU with
``torchrun --nproc-per-node N -m hcmoco_amd.pycontrast.main_contrast <flags>`` (or ``srun`` as
the reference scripts do; SLURM variables are honoured), or run it single-process.
"""
import os
import sys

if __package__ in (None, ''):        # executed as a script: make the package importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    __package__ = 'hcmoco_amd.pycontrast'

import torch

from .options.train_options import TrainOptions
from .learning.contrast_trainer import ContrastTrainer
from .networks.build_backbone import build_model
from .memory.build_memory import build_mem
from .datasets.synthetic import build_synthetic_contrast_loader


def main(argv=None, engine=None):
    args = TrainOptions().parse(argv)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    ngpus_per_node = max(1, torch.cuda.device_count())
    return main_worker(0, ngpus_per_node, args, engine)


def main_worker(gpu, ngpus_per_node, args, engine=None):
    trainer = ContrastTrainer(args, engine=engine)
    trainer.init_ddp_environment(gpu, ngpus_per_node)
    args.channels_last = bool(getattr(args, 'channels_last', False))

    model, model_ema = build_model(args)

    if args.synthetic:
        train_dataset, train_loader, train_sampler = build_synthetic_contrast_loader(
            args, trainer.device, args.rank, args.world_size)
    else:       # image files -> the positional tuple of SURVEY appendix B (datasets/ntu_mpii.py, PIL + numpy)
        from .datasets.util import build_own_contrast_loader
        train_dataset, train_loader, train_sampler = build_own_contrast_loader(
            args, args.rank, args.world_size, ngpus_per_node)

    contrast = build_mem(args, len(train_dataset))
    contrast.to(trainer.device)

    if args.pretrain is not None:               # stage-1 -> stage-2 hand-off (main_contrast.py:52-67)
        ckpt = torch.load(args.pretrain, map_location='cpu')
        converted = {k[7:]: v for k, v in ckpt['model'].items()}
        own = model.state_dict()
        unmatched = [k for k in own if k not in converted]
        own.update({k: v for k, v in converted.items() if k in own})
        print('Unmatched Keys: {}'.format(', '.join(unmatched)))
        model.load_state_dict(own)
        contrast.load_state_dict(ckpt['contrast'])

    # the cross-entropy criteria of the reference (:70-81) live inside the fused kernels here
    criterion = None
    model.to(trainer.device)
    # torch.optim.SGD as in the reference (:78-81); on the GPU its `fused` implementation updates all
    # ~1000 parameter tensors in a handful of launches instead of one foreach chain per operation
    optimizer = torch.optim.SGD(model.parameters(), lr=args.learning_rate, momentum=args.momentum,
                                weight_decay=args.weight_decay, fused=trainer.device.type == 'cuda')

    model, model_ema, optimizer = trainer.wrap_up(model, model_ema, optimizer)
    trainer.broadcast_memory(contrast)
    start_epoch = trainer.resume_model(model, model_ema, contrast, optimizer)
    trainer.init_tensorboard_logger()

    outs = None
    for epoch in range(start_epoch, args.epochs + 1):
        train_sampler.set_epoch(epoch)
        trainer.adjust_learning_rate(optimizer, epoch)
        outs = trainer.train(epoch, train_loader, model, model_ema, contrast, criterion, optimizer)
        trainer.logging(epoch, outs, optimizer.param_groups[0]['lr'])
        trainer.save(model, model_ema, contrast, optimizer, epoch)
    return outs, trainer, model, contrast


if __name__ == '__main__':
    main()
