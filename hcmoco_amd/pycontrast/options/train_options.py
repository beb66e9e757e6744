"""Training flags and derived fields (reference: options/train_options.py:9-73)."""
import math
import os

from .base_options import BaseOptions


class TrainOptions(BaseOptions):

    def flags(self):
        return super().flags() + [
            (('--aug',), dict(type=str, default='A')),
            (('--beta',), dict(type=float, default=0.5)),
            (('--warm',), dict(action='store_true')),
            (('--amp',), dict(action='store_true')),
            (('--opt_level',), dict(type=str, default='O2', choices=['O1', 'O2'])),
            (('--n_class',), dict(type=int, default=31)),
        ]

    def modify_options(self, opt):
        opt = self.override_options(opt)
        opt.lr_decay_epochs = [int(t) for t in opt.lr_decay_epochs.split(',')]
        opt.in_channel_list = [int(t) for t in opt.in_channel_list.split(',')]

        # run name, suffix by suffix as the reference builds it (:39-47)
        name = '_'.join(str(v) for v in (opt.method, opt.arch, opt.modal, 'Jig', opt.jigsaw, opt.mem,
                                         'aug', opt.aug, opt.head, opt.nce_t, opt.tag))
        if opt.amp:
            name += '_amp_' + opt.opt_level
            # the reference's apex amp is fp16 (learning/contrast_trainer.py:65-72); this build's mixed precision is
            # bf16 autocast of the encoders: same flag, no loss scaling needed
            opt.encoder_dtype = 'bf16'
        if opt.cosine:
            name += '_cosine'

        # large-batch warm-up (:50-63)
        opt.warm = opt.warm or opt.batch_size > 256
        if opt.warm:
            name += '_warm'
            opt.warmup_from = 0.01
            opt.warm_epochs = 10 if opt.epochs > 500 else 5
            if opt.cosine:
                eta_min = opt.learning_rate * (opt.lr_decay_rate ** 3)
                opt.warmup_to = eta_min + (opt.learning_rate - eta_min) * (
                    1 + math.cos(math.pi * opt.warm_epochs / opt.epochs)) / 2
            else:
                opt.warmup_to = opt.learning_rate
        opt.model_name = name

        # folders are created at parse time, like the reference does (:66-71)
        opt.model_folder = os.path.join(opt.model_path, opt.model_name)
        opt.tb_folder = os.path.join(opt.tb_path, opt.model_name)
        for folder in (opt.model_folder, opt.tb_folder):
            os.makedirs(folder, exist_ok=True)
        return opt
