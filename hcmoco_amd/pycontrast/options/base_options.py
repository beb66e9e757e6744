"""Command-line surface of the pre-training entry point.

Mirrors the flag set and the method presets of the reference
(/root/reference/pycontrast/options/base_options.py:12-22, :31-151, :168-194): same flag names,
types, defaults and derived fields, so existing launch scripts keep working.  Differences, all
additive:
  * ``--method CMCJointsPri3DRGBD2S`` is accepted (the released second-stage scripts pass it but
    the reference's ``choices`` list rejects it);
  * ``--mem`` accepts every ``bank*`` preset value, not only ``bank``/``moco``;
  * ``--synthetic*`` flags select the on-device synthetic batch source used for benchmarking.
"""
import argparse

# method -> (modal, jigsaw, mem, aug, head, nce_t)          base_options.py:12-22
METHOD_PRESETS = {
    'InsDis':               ('RGB',    False, 'bank',             'A', 'linear', 0.07),
    'CMC':                  ('CMC',    False, 'bank',             'C', 'linear', 0.07),
    'MoCo':                 ('RGB',    False, 'moco',             'A', 'linear', 0.07),
    'PIRL':                 ('RGB',    True,  'bank',             'A', 'linear', 0.07),
    'MoCov2':               ('RGB',    False, 'moco',             'B', 'mlp',    0.2),
    'CMCv2':                ('CMC',    False, 'moco',             'E', 'mlp',    0.2),
    'InfoMin':              ('RGB',    True,  'moco',             'D', 'mlp',    0.15),
    'CMCRGBD2S':            ('RGBD2S', False, 'bank',             'C', 'linear', 0.07),
    'CMCJointsPri3DRGBD2S': ('RGBD2S', False, 'bank+jointspri3d', 'C', 'linear', 0.07),
}

_S, _I, _F = str, int, float
# (flags, kwargs) in the reference's order
BASE_FLAGS = [
    (('--data_folder',), dict(type=_S, default='./data')),
    (('--train_file_list',), dict(type=_S, default='')),
    (('--val_file_list',), dict(type=_S, default='')),
    (('--model_path',), dict(type=_S, default='./save')),
    (('--tb_path',), dict(type=_S, default='./tb')),
    (('--pretrain',), dict(type=_S, default=None)),
    (('--tag',), dict(type=_S, default='')),
    (('--print_freq',), dict(type=_I, default=10)),
    (('--save_freq',), dict(type=_I, default=20)),
    (('--batch_size',), dict(type=_I, default=256)),
    (('-j', '--num_workers'), dict(type=_I, default=40)),
    (('--epochs',), dict(type=_I, default=200)),
    (('--learning_rate',), dict(type=_F, default=0.03)),
    (('--lr_decay_epochs',), dict(type=_S, default='120,160')),
    (('--lr_decay_rate',), dict(type=_F, default=0.1)),
    (('--weight_decay',), dict(type=_F, default=1e-4)),
    (('--momentum',), dict(type=_F, default=0.9)),
    (('--cosine',), dict(action='store_true')),
    (('--downstream_training',), dict(action='store_true')),
    (('--method',), dict(type=_S, default='Customize', choices=list(METHOD_PRESETS) + ['Customize'])),
    (('--modal',), dict(type=_S, default='RGB', choices=['RGB', 'CMC', 'RGBD2S'])),
    (('--in_channel_list',), dict(type=_S, default='1,2')),
    (('--linear_feat_map',), dict(type=_I, default=0)),
    (('--width',), dict(type=_I, default=18)),
    (('--dataset',), dict(type=_S, default='')),
    (('--IN_Pretrain',), dict(type=_S, default=None)),
    (('--pri3d_num_samples_per_image',), dict(type=_I, default=400)),
    (('--modality_missing',), dict(type=_I, default=0)),
    (('--mpii_root',), dict(type=_S, default='')),
    (('--pool_method',), dict(type=_S, default='mean')),
    (('--depth_Pretrain',), dict(type=_S, default=None)),
    (('--cmc_loss_weight',), dict(type=_F, default=1.0)),
    (('--skeleton_meta_name',), dict(type=_S, default='mpii')),
    (('--coco_root',), dict(type=_S, default='')),
    (('--not_use_weighted_sampler',), dict(action='store_true', default=False)),
    (('--seg_root',), dict(type=_S, default='')),
    (('--seg_file_list',), dict(type=_S, default='')),
    (('--seg_val_file_list',), dict(type=_S, default='')),
    (('--mask_seg_depth',), dict(action='store_true', default=False)),
    (('--test_type',), dict(type=_I, default=0)),
    (('--cmc_loss_weights',), dict(type=_F, default=1)),
    (('--other_loss_weights',), dict(type=_F, default=1)),
    (('--supervise_type',), dict(type=_I, default=0)),
    (('--mask_seg_rgb',), dict(action='store_true', default=False)),
    (('--temperature',), dict(type=_F, default=0.07)),
    (('--random_flip',), dict(type=_I, default=0)),
    (('--jigsaw',), dict(action='store_true')),
    (('--mem',), dict(type=_S, default='bank',
                      choices=sorted({p[2] for p in METHOD_PRESETS.values()}))),
    (('--arch',), dict(type=_S, default='resnet50')),
    (('-d', '--feat_dim'), dict(type=_I, default=128)),
    (('-k', '--nce_k'), dict(type=_I, default=65536)),
    (('-m', '--nce_m'), dict(type=_F, default=0.5)),
    (('-t', '--nce_t'), dict(type=_F, default=0.07)),
    (('--alpha',), dict(type=_F, default=0.999)),
    (('--head',), dict(type=_S, default='linear', choices=['linear', 'mlp'])),
    (('--resume',), dict(type=_S, default='', metavar='PATH')),
    (('--world-size',), dict(type=_I, default=-1)),
    (('--rank',), dict(type=_I, default=-1)),
    (('--dist-url',), dict(type=_S, default='tcp://127.0.0.1:23456')),
    (('--dist-backend',), dict(type=_S, default='nccl')),
    (('--seed',), dict(type=_I, default=None)),
    (('--gpu',), dict(type=_I, default=None)),
    (('--multiprocessing-distributed',), dict(action='store_true')),
    # ---- additions of this build (not in the reference) ----
    (('--synthetic',), dict(action='store_true',
                            help='draw batches from the on-device synthetic source (SURVEY 8d)')),
    (('--synthetic_n_data',), dict(type=_I, default=131072)),
    (('--synthetic_size',), dict(type=_I, default=256)),
    (('--synthetic_steps',), dict(type=_I, default=50, help='batches per epoch in synthetic mode')),
    (('--image_size',), dict(type=_I, default=320, help='side of the square crops of the image datasets (the '
                                                        'reference hard-codes its class default, 320)')),
    (('--synthetic_ntu',), dict(type=_I, default=0,
                                help='synthetic batches carry the NTU-only items 9-15 (use_rgb at position 11)')),
    (('--synthetic_p_rgb',), dict(type=float, default=1.0, help='P(use_rgb = 1) of NTU-style synthetic samples')),
    (('--sampled_projection',), dict(type=_I, default=1,
                                     help='1: apply merge_all_res + the 1x1 feature-map projection only at the '
                                          'pixels the losses sample (same math, SURVEY 8f-1); 0: full maps')),
    (('--wgrad_stream',), dict(type=_I, default=8,
                               help='n > 0: the encoder programs hand their weight gradients (own kernels '
                                    'and MIOpen\'s five-launch path alike) to a side stream, n layers at a time, so that the reverse '
                                    'chain dx -> BatchNorm -> dx is not queued behind them (csrc/torch_glue); 0: in line')),
    # escape hatches of the ROCm runtime (reachable from the CLI, not only from test code).  All default to on;
    # 0 gives the plain-autograd twin the parity tests compare with (tests/test_trainer_gpu.py, tests/test_section_gpu.py)
    (('--fused_section',), dict(type=_I, default=1,
                                help='1: pooling, heads, all-gather, bank NCE + update, pixel sampling, sampled projection and '
                                     'the three feature-map losses as ONE autograd node (csrc/section.hip); 0: module by module')),
    (('--async_wgrad',), dict(type=_I, default=1,
                              help='1: the encoder programs\' reverse loops and weight gradients are issued by helper threads '
                                   'while the autograd thread walks on (csrc/torch_glue); 0: everything on the autograd thread')),
    (('--flat_sgd',), dict(type=_I, default=1,
                           help='1: one SGD launch per encoder over its flat parameter buffer (learning/flat_sgd.py; '
                                'checkpoints keep the per-parameter layout); 0: torch.optim.SGD as it is')),
    (('--grad_sync',), dict(type=_S, default='auto', choices=['auto', 'ddp', 'flat', 'overlap'],
                            help='N>1 gradient averaging: ddp = DistributedDataParallel; overlap = in-place RCCL '
                                 'all-reduces of the encoders\' flat gradient buffers, launched chunk by chunk while '
                                 'backward is still running (learning/grad_sync.py); flat = ONE all-reduce after '
                                 'backward; auto = overlap on ROCm, ddp on CPU')),
    (('--fmap_dtype',), dict(type=_S, default='fp32', choices=['fp32', 'bf16', 'fp32_exact'],
                             help='arithmetic of the dense / SCL feature-map contractions: fp32 = fp32-ACCURATE products on the '
                                  'bf16 matrix cores (operands split into two bf16 pieces, 3 / 4 MFMA terms per product, fp32 '
                                  'accumulation; 4e-6..6e-6 of float64, inside the 1e-5 / 1e-4 parity gate) or bf16 = operands '
                                  'rounded to bf16, fp32 accumulation (BASELINE config 5); fp32_exact = three bf16 pieces per operand, all nine '
                                  'piece products: an exact-product fp32 contraction at 3 x the matrix work (not a speed mode)')),
    (('--bank_dtype',), dict(type=_S, default='fp32', choices=['fp32', 'bf16'],
                             help='storage type of the memory banks (bf16: BASELINE config 5)')),
    (('--encoder_dtype',), dict(type=_S, default='fp32', choices=['fp32', 'bf16'],
                                help='arithmetic of the HRNet encoders: fp32 (the reference) or bf16 mixed precision -- '
                                     'bf16 convolutions under torch.autocast, fp32 batch-norm statistics, fp32 master '
                                     'weights and optimizer, fp32 loss section (BASELINE config 5).  --amp (the '
                                     'reference\'s apex fp16 switch, train_options.py:16-19) selects it too')),
]


class BaseOptions(object):
    """``parse()`` -> argparse.Namespace, printed like the reference does (:153-163)."""

    override_dict = {k: list(v) for k, v in METHOD_PRESETS.items()}

    def __init__(self):
        self.parser = None
        self.opt = None

    def flags(self):
        return list(BASE_FLAGS)

    def initialize(self, parser):
        for names, kw in self.flags():
            parser.add_argument(*names, **kw)
        return parser

    def override_options(self, opt):
        preset = METHOD_PRESETS.get(opt.method)
        if preset is not None:
            opt.modal, opt.jigsaw, opt.mem, opt.aug, opt.head, opt.nce_t = preset
        return opt

    def modify_options(self, opt):
        raise NotImplementedError

    def print_options(self, opt):
        lines = ['----------------- Options ---------------']
        for key in sorted(vars(opt)):
            val, dflt = getattr(opt, key), self.parser.get_default(key)
            note = '' if val == dflt else '\t[default: %s]' % str(dflt)
            lines.append('{:>35}: {:<30}{}'.format(str(key), str(val), note))
        lines.append('----------------- End -------------------')
        print('\n'.join(lines))

    def parse(self, argv=None):
        if self.parser is None:
            self.parser = self.initialize(argparse.ArgumentParser('arguments options'))
        opt = self.modify_options(self.parser.parse_args(argv))
        self.opt = opt
        self.print_options(opt)
        return opt
