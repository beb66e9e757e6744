"""HRNetV2 backbone (w18 / w32 / w48) returning the four multi-resolution maps.

Encoder plugin of the pre-training path; runs on stock PyTorch-ROCm (MIOpen convolutions) -- the
hand-written HIP kernels of this repo start where these maps are consumed.  Topology, parameter
names (``state_dict`` keys) and initialisation follow the reference
(/root/reference/pycontrast/networks/official_hrnet/official_hrnet.py:247-454 and its three yaml
files), so ImageNet / first-stage checkpoints load unchanged.  The topology is a Python table
here instead of a yacs config read from a cwd-relative yaml path (official_hrnet.py:498-503).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOMENTUM = 0.01                                           # official_hrnet.py:23
# Module attributes, not environment switches (r05): the parity tests flip them to build the plain-ATen / module-path
# twin of a model inside one process (tests/test_glue_gpu.py, test_trainer_gpu.py, test_exact_gpu.py).
FUSED_BN = True          # hcm_bn_act_* on the GPU (False: stock ops)
CONV_GLUE = True         # torch.ops.hcmoco.conv2d (False: ATen)
ENCODER_PROGRAM = True   # whole encoder as one C++-executed program
# HRNet branch i on HIP stream i of the encoder.  Off: measured 530 vs 572 samples/s -- the ~300 cross-stream
# event waits per pass cost more than the extra overlap buys (134 ms/step with GPU_MAX_HW_QUEUES=8).
BRANCH_STREAMS = False
FUSE_UPSAMPLE_ADD = True  # fuse-layer terms as one instruction


def bn_act_supported(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and (x.shape[2] * x.shape[3]) % 4 == 0)


def upsample_bilinear(x, size):
    """``F.interpolate(x, size, mode='bilinear')`` (align_corners=False).  On the MI355X the forward
    runs in ``hcm_upsample_bilinear2d`` (ATen's NCHW forward is the single largest kernel of the
    step there, profiles/r01_bench_one_step_summary.csv); CPU tensors use ATen."""
    if x.is_cuda and x.dtype == torch.float32:
        if x.is_contiguous():
            return _glue_op('upsample_bilinear')(x, int(size[0]), int(size[1]))
        from ... import hip_ops
        return hip_ops.upsample_bilinear(x, size)
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)

# stage -> (modules, blocks per branch, block type); branch widths = width * 2**i
STAGES = {
    'stage1': dict(modules=1, branches=1, blocks=4, kind='bottleneck', planes=64),
    'stage2': dict(modules=1, branches=2, blocks=4, kind='basic'),
    'stage3': dict(modules=4, branches=3, blocks=4, kind='basic'),
    'stage4': dict(modules=3, branches=4, blocks=4, kind='basic'),
}


_GLUE_OPS = {}


def _glue_op(name):
    """torch.ops.hcmoco.<name>.default, resolved once (the packet lookup costs microseconds per call)."""
    op = _GLUE_OPS.get(name)
    if op is None:
        from ... import _lib
        op = _GLUE_OPS[name] = getattr(_lib.torch_glue(), name).default
    return op


class Conv2d(nn.Conv2d):
    """nn.Conv2d (bias-free, groups = dilation = 1) whose ROCm path is ``torch.ops.hcmoco.conv2d``:
    the same MIOpen kernels ATen would pick, issued from one C++ autograd node with cached
    descriptors/algorithms (ATen spends ~125 us of host time per layer and step on a 25 us kernel;
    csrc/torch_glue/hcm_torch_glue.cpp).  HCM_CONV_GLUE=0 or any other configuration: stock ATen."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.glue_ok = (self.bias is None and self.groups == 1 and self.dilation == (1, 1)
                        and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1]
                        and isinstance(self.padding, tuple) and self.padding_mode == 'zeros')

    def forward(self, x):
        if CONV_GLUE and self.glue_ok and x.is_cuda and x.dtype == torch.float32:
            return _glue_op('conv2d')(x, self.weight, self.stride[0], self.padding[0])
        return super().forward(x)


def conv_bn(conv, bn, x, residual=None, relu=False):
    """``relu?(bn(conv(x)) + residual?)``; a training step on the MI355X runs it as ONE autograd node
    (torch.ops.hcmoco.conv_bn_act: MIOpen convolution + hcm_bn_act_*), anything else as the stock
    composition of the two modules."""
    if (CONV_GLUE and FUSED_BN and bn.training and conv.glue_ok and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4):
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        ho, wo = (x.shape[2] + 2 * p - k) // s + 1, (x.shape[3] + 2 * p - conv.kernel_size[1]) // s + 1
        if (ho * wo) % 4 == 0:
            return _glue_op('conv_bn_act')(x, conv.weight, s, p, residual, bn.weight, bn.bias, bn.running_mean,
                                           bn.running_var, bn.momentum, bn.eps, relu)
    return bn(conv(x), residual=residual, relu=relu)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d minus its per-layer ``num_batches_tracked += 1`` launch: with a fixed momentum
    the counter never enters the arithmetic, and an HRNet-w18 pair has 618 BN layers (one 5 us kernel
    each per step).  ``HighResolutionNet.forward`` advances all counters with ONE foreach launch, so
    the buffers (and checkpoints) evolve exactly as in the reference."""

    def forward(self, x, residual=None, relu=False):
        """``relu?(bn(x) + residual?)``.  Training steps on the MI355X run it as ONE fused op
        (``hcm_bn_act_forward/backward``, a (channel, slice) grid instead of the library's one
        workgroup per channel); everything else is the stock composition."""
        if FUSED_BN and self.training and bn_act_supported(x):
            return _glue_op('bn_act')(x, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                      self.momentum, self.eps, relu)
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias,
                         self.training or not self.track_running_stats, self.momentum, self.eps)
        if residual is not None:
            y = y + residual
        return F.relu(y, inplace=True) if relu else y


def _bn(c):
    return BatchNorm2d(c, momentum=BN_MOMENTUM)


class ConvBn(nn.Sequential):
    """conv -> bn [-> relu] with the reference's Sequential child names ('0', '1', ['2']); the
    normalisation, the optional residual and the ReLU run as one op."""

    def forward(self, x, residual=None):
        return conv_bn(self[0], self[1], x, residual, len(self) == 3)

    def emit(self, pb, a, res=-1):
        return pb.conv_bn(self[0], self[1], a, res, len(self) == 3)


def _conv_bn(cin, cout, k, stride=1, relu=False):
    layers = [Conv2d(cin, cout, k, stride, k // 2, bias=False), _bn(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return ConvBn(*layers)


class ProgramBuilder(object):
    """Flattens the module tree into the instruction list ``torch.ops.hcmoco.run_encoder`` executes
    (csrc/torch_glue: one autograd node for the whole encoder, C++ forward loop, reverse loop that can run
    on a helper thread).  Instructions are 12 ints ``op dst a b layer stride pad relu out_h out_w 0 0``
    over value slots; slot 0 is the input image.  ``ok`` turns False when a layer falls outside what the
    kernels cover (then the module-by-module path runs)."""
    CONV_BN, ADD, RELU, UPSAMPLE, UPSAMPLE_ADD = 0, 1, 2, 3, 4

    def __init__(self, in_shape):
        self.instr, self.params, self.buffers = [], [], []
        self.shapes = [tuple(in_shape)]          # (C, H, W) per slot
        self.ok = True
        self.sid = 0                             # stream id stamped on the next instructions (0 = caller's stream)

    def on(self, sid):
        """Following instructions run on HIP stream ``sid`` of the encoder (0..3; the executor inserts
        the cross-stream waits).  Branch i of every HighResolutionModule lives on stream i."""
        self.sid = sid if BRANCH_STREAMS else 0

    def _new(self, shape):
        self.shapes.append(tuple(shape))
        return len(self.shapes) - 1

    def conv_bn(self, conv, bn, a, res=-1, relu=False):
        _, h, w = self.shapes[a]
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - conv.kernel_size[1]) // s + 1
        if not conv.glue_ok or (ho * wo) % 4 != 0 or not bn.affine or not bn.track_running_stats:
            self.ok = False
        layer = len(self.params) // 3
        self.params += [conv.weight, bn.weight, bn.bias]
        self.buffers += [bn.running_mean, bn.running_var]
        dst = self._new((conv.out_channels, ho, wo))
        self.instr += [self.CONV_BN, dst, a, res, layer, s, p, int(relu), 0, 0, self.sid, 0]
        return dst

    def add(self, a, b):
        dst = self._new(self.shapes[a])
        self.instr += [self.ADD, dst, a, b, 0, 0, 0, 0, 0, 0, self.sid, 0]
        return dst

    def relu(self, a):
        dst = self._new(self.shapes[a])
        self.instr += [self.RELU, dst, a, -1, 0, 0, 0, 0, 0, 0, self.sid, 0]
        return dst

    def upsample(self, a, size):
        dst = self._new((self.shapes[a][0], int(size[0]), int(size[1])))
        self.instr += [self.UPSAMPLE, dst, a, -1, 0, 0, 0, 0, int(size[0]), int(size[1]), self.sid, 0]
        return dst

    def upsample_add(self, a, acc, relu=False):
        """relu?(acc + upsample(a)) to acc's size: a term of a fuse layer (and its ReLU when it is the last) as one
        instruction instead of upsample + add (+ relu)."""
        dst = self._new(self.shapes[acc])
        _, h, w = self.shapes[acc]
        self.instr += [self.UPSAMPLE_ADD, dst, a, acc, 0, 0, 0, int(relu), int(h), int(w), self.sid, 0]
        return dst

    def chain(self, seq, a):
        for m in seq:
            a = m.emit(self, a)
        return a


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = conv_bn(self.conv1, self.bn1, x, None, True)
        return conv_bn(self.conv2, self.bn2, y, skip, True)

    def emit(self, pb, a):
        skip = a if self.downsample is None else self.downsample.emit(pb, a)
        y = pb.conv_bn(self.conv1, self.bn1, a, -1, True)
        return pb.conv_bn(self.conv2, self.bn2, y, skip, True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = conv_bn(self.conv1, self.bn1, x, None, True)
        y = conv_bn(self.conv2, self.bn2, y, None, True)
        return conv_bn(self.conv3, self.bn3, y, skip, True)

    def emit(self, pb, a):
        skip = a if self.downsample is None else self.downsample.emit(pb, a)
        y = pb.conv_bn(self.conv1, self.bn1, a, -1, True)
        y = pb.conv_bn(self.conv2, self.bn2, y, -1, True)
        return pb.conv_bn(self.conv3, self.bn3, y, skip, True)


def _block_chain(block, cin, planes, n):
    down = None
    if cin != planes * block.expansion:
        down = ConvBn(Conv2d(cin, planes * block.expansion, 1, bias=False), _bn(planes * block.expansion))
    layers = [block(cin, planes, 1, down)]
    layers += [block(planes * block.expansion, planes) for _ in range(n - 1)]
    return nn.Sequential(*layers)


class HighResolutionModule(nn.Module):
    """Parallel branches followed by full cross-resolution fusion (official_hrnet.py:108-244)."""

    def __init__(self, widths, blocks):
        super().__init__()
        nb = len(widths)
        self.num_branches = nb
        self.branches = nn.ModuleList([_block_chain(BasicBlock, w, w, blocks) for w in widths])
        self.relu = nn.ReLU(inplace=True)
        if nb == 1:
            self.fuse_layers = None
            return
        rows = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:                      # coarser -> finer: 1x1 conv, upsampled in forward
                    row.append(_conv_bn(widths[j], widths[i], 1))
                elif j == i:
                    row.append(None)
                else:                          # finer -> coarser: (i-j) stride-2 3x3 convs
                    steps = [_conv_bn(widths[j], widths[j], 3, 2, relu=True) for _ in range(i - j - 1)]
                    steps.append(_conv_bn(widths[j], widths[i], 3, 2))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs):
        xs = [br(x) for br, x in zip(self.branches, xs)]
        if self.num_branches == 1:
            return xs
        outs = []
        for i, row in enumerate(self.fuse_layers):
            y = xs[0] if i == 0 else row[0](xs[0])
            for j in range(1, self.num_branches):
                if j == i:
                    y = y + xs[j]
                elif j > i:
                    y = y + upsample_bilinear(row[j](xs[j]), xs[i].shape[-2:])
                else:                          # the last conv-bn of the chain adds the running sum itself
                    t = xs[j]
                    for step in row[j][:-1]:
                        t = step(t)
                    y = row[j][-1](t, residual=y)
            outs.append(self.relu(y))
        return outs

    def emit(self, pb, xs):
        """Same dataflow as forward(), as instructions."""
        ys = []
        for j, (br, a) in enumerate(zip(self.branches, xs)):
            pb.on(j)
            ys.append(pb.chain(br, a))
        xs = ys
        if self.num_branches == 1:
            return xs
        outs = []
        for i, row in enumerate(self.fuse_layers):
            pb.on(i)
            y = xs[0] if i == 0 else pb.chain(row[0], xs[0])
            last = self.num_branches - 1
            done = False                       # the output's ReLU went into the last term's instruction
            for j in range(1, self.num_branches):
                if j == i:
                    y = pb.add(y, xs[j])
                elif j > i:
                    if FUSE_UPSAMPLE_ADD:      # same sum in the same order, one launch per term (two or three before)
                        y = pb.upsample_add(row[j].emit(pb, xs[j]), y, relu=(j == last))
                        done = j == last
                    else:
                        y = pb.add(y, pb.upsample(row[j].emit(pb, xs[j]), pb.shapes[xs[i]][1:]))
                else:
                    t = xs[j]
                    for step in row[j][:-1]:
                        t = step.emit(pb, t)
                    y = row[j][-1].emit(pb, t, y)
            outs.append(y if done else pb.relu(y))
        return outs


class HighResolutionNet(nn.Module):
    _next_tag = 0

    def __init__(self, width=18):
        super().__init__()
        self.width = width
        self.conv1 = Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = _bn(64)
        self.conv2 = Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = _bn(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = _block_chain(Bottleneck, 64, 64, 4)
        prev = [256]
        for s in (2, 3, 4):
            cfg = STAGES['stage%d' % s]
            widths = [width * 2 ** i for i in range(cfg['branches'])]
            setattr(self, 'transition%d' % (s - 1), self._transition(prev, widths))
            setattr(self, 'stage%d' % s, nn.Sequential(*[HighResolutionModule(widths, cfg['blocks'])
                                                         for _ in range(cfg['modules'])]))
            prev = widths
        self.out_channels = prev
        self.init_weights()
        self._counters = None
        self._programs = {}
        # id under which the encoder runtime publishes this network's flat gradient buffer
        # (torch.ops.hcmoco.grad_chunk_wait, learning/grad_sync.py); 0 = not published
        HighResolutionNet._next_tag += 1
        self.grad_tag = HighResolutionNet._next_tag
        self.last_program = None        # ProgramBuilder of the most recent forward that ran as a program

    def _count_batch(self):
        if self._counters is None or self._counters[0].device != self.conv1.weight.device:
            self._counters = [m.num_batches_tracked for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        with torch.no_grad():
            torch._foreach_add_(self._counters, 1)

    @staticmethod
    def _transition(prev, cur):
        layers = []
        for i, c in enumerate(cur):
            if i < len(prev):
                layers.append(None if prev[i] == c else _conv_bn(prev[i], c, 3, relu=True))
            else:    # new, coarser branch: stride-2 convs from the last previous branch
                steps = []
                for k in range(i + 1 - len(prev)):
                    last = k == i - len(prev)
                    steps.append(_conv_bn(prev[-1], c if last else prev[-1], 3, 2, relu=True))
                layers.append(nn.Sequential(*steps))
        return nn.ModuleList(layers)

    def init_weights(self):
        """normal(std=0.001) convs, unit BN (official_hrnet.py:456-463)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def program(self, in_shape):
        """(builder, output slots) for a (C, H, W) input, compiled once."""
        key = tuple(in_shape)
        hit = self._programs.get(key)
        if hit is None:
            pb = ProgramBuilder(key)
            a = pb.conv_bn(self.conv1, self.bn1, 0, -1, True)
            a = pb.conv_bn(self.conv2, self.bn2, a, -1, True)
            ys = [pb.chain(self.layer1, a)]
            for s in (2, 3, 4):
                trans = getattr(self, 'transition%d' % (s - 1))
                xs = []
                for i, t in enumerate(trans):
                    if t is None:
                        xs.append(ys[i])
                    else:
                        src = ys[-1] if (s > 2 or len(ys) == 1) else ys[i]
                        pb.on(i)
                        xs.append(t.emit(pb, src) if isinstance(t, ConvBn) else pb.chain(t, src))
                for mod in getattr(self, 'stage%d' % s):
                    xs = mod.emit(pb, xs)
                ys = xs
            hit = self._programs[key] = (pb, ys)
        return hit

    def _program_args(self, x):
        """Arguments of torch.ops.hcmoco.run_encoder, or None when this call must take the module path
        (CPU, eval mode, switched off, a layer the kernels do not cover)."""
        if not (ENCODER_PROGRAM and CONV_GLUE and FUSED_BN and self.training and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 4):
            return None
        pb, outs = self.program(x.shape[1:])
        if not pb.ok:
            self.last_program = None
            return None
        self.last_program = pb
        return (x, pb.params, pb.buffers, pb.instr, outs, len(pb.shapes), self.bn1.momentum, self.bn1.eps,
                self.grad_tag)

    def forward_async(self, x, node_at_wait=False):
        """Start the forward on the helper thread of the CURRENT stream (C++, no GIL); returns a handle
        for forward_wait, or None when the module path has to run (then call forward).  ``node_at_wait``: the autograd node
        is made by ``forward_wait`` on the calling thread instead of on the helper thread (csrc/torch_glue)."""
        args = self._program_args(x)
        if args is None:
            return None
        self._count_batch()
        return _glue_op('encoder_forward_async')(*args, node_at_wait)

    @staticmethod
    def forward_wait(handle):
        return list(_glue_op('encoder_forward_wait')(handle))

    def forward(self, x):
        args = self._program_args(x)
        if args is not None:
            self._count_batch()
            return list(_glue_op('run_encoder')(*args))
        self.last_program = None
        if self.training:
            self._count_batch()
        x = conv_bn(self.conv1, self.bn1, x, None, True)
        x = conv_bn(self.conv2, self.bn2, x, None, True)
        x = self.layer1(x)
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(self, 'transition%d' % (s - 1))
            xs = []
            for i, t in enumerate(trans):
                if t is None:
                    xs.append(ys[i])
                else:   # stage 2 branches all start from the stem; later new branches from the coarsest map
                    xs.append(t(ys[-1] if (s > 2 or len(ys) == 1) else ys[i]))
            ys = getattr(self, 'stage%d' % s)(xs)
        return ys


def get_hrnet_w18_backbone():
    return HighResolutionNet(18)


def get_hrnet_w32_backbone():
    return HighResolutionNet(32)


def get_hrnet_w48_backbone():
    return HighResolutionNet(48)
