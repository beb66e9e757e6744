"""Conv/BN building blocks with the reference's module names, so ``state_dict`` keys match
(/root/reference/pycontrast/networks/pointnet2/pytorch_utils.py:5-200):
``...layer{i}.conv.weight``, ``...layer{i}.bn.bn.weight`` etc."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

FUSED = True      # module attribute: tests flip it to get the stock-op twin of a layer


class _BN(nn.Sequential):
    def __init__(self, norm, channels):
        super().__init__()
        self.add_module('bn', norm(channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class _ConvBNAct(nn.Sequential):
    """conv (kaiming-normal, bias only without BN) -> BN -> activation (post-activation form)."""

    def __init__(self, conv, norm, cin, cout, bn, activation):
        super().__init__()
        unit = conv(cin, cout, kernel_size=1, stride=1, padding=0, bias=not bn)
        nn.init.kaiming_normal_(unit.weight)
        if not bn:
            nn.init.constant_(unit.bias, 0)
        self.add_module('conv', unit)
        if bn:
            self.add_module('bn', _BN(norm, cout))
        if activation is not None:
            self.add_module('activation', activation)
        self._fusable = (conv is nn.Conv2d and bn and norm is nn.BatchNorm2d
                         and (activation is None or isinstance(activation, nn.ReLU)))

    def forward(self, x):
        """1x1 conv -> BatchNorm2d -> ReLU of the shared MLPs.  A training step on the MI355X runs the
        three as ONE autograd node (torch.ops.hcmoco.conv_bn_act: MIOpen convolution + hcm_bn_act_*;
        MIOpen's own batch-norm takes 0.5 ms per layer on the [32,64,1024,32] ball tensors)."""
        if (self._fusable and FUSED and self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and (x.shape[2] * x.shape[3]) % 4 == 0):
            from .... import _lib
            bn = self.bn.bn
            bn.num_batches_tracked.add_(1)          # nn.BatchNorm2d.forward's bookkeeping
            return _lib.torch_glue().conv_bn_act(x, self.conv.weight, 1, 0, None, bn.weight, bn.bias, bn.running_mean,
                                                 bn.running_var, bn.momentum, bn.eps, hasattr(self, 'activation'))
        return super().forward(x)


    def forward_grouped(self, offsets, features, idx):
        """This layer applied to QueryAndGroup's output [xyz[idx] - centre ; features[idx]] without building it (r05): the
        FEATURE half of a 1x1 convolution commutes with the gather, so the N source points are projected once -- P = W_f
        features, a GEMM nsample times smaller than the convolution over the grouped tensor -- and normalisation + ReLU
        read z = P[idx] + W_xyz offsets through the gather (``pointnet2_hip.ball_project``).  The COORDINATE half is
        evaluated on the relative offsets themselves (``offsets`` [B, 3, np, ns] = xyz[idx] - centre, the reference's
        grouped_xyz: geometry, made ahead of the feature path): r05 commuted it too (P' = W [xyz ; features] minus
        Q = W_xyz centre), which subtracts two numbers of the cloud's extent to get one of the ball's radius -- 40 x the
        reference's round-off on the 2.5 cm balls (r06, tests/test_pn_reference_gpu.py).
        features [B, C, N] or None, idx [B, np, ns] -> [B, C1, np, ns].
        Reference: pointnet2_utils.py:231-268 + pytorch_utils.py:5-33."""
        from .... import pointnet2_hip
        bn = self.bn.bn
        W = self.conv.weight.view(self.conv.weight.shape[0], -1)            # [C1, 3 + C], xyz columns first
        P = None
        if features is not None:
            Wf = W[:, 3:].contiguous()
            if pointnet2_hip.point_project_supported(Wf.shape[0], Wf.shape[1], features.shape[2]):
                P = pointnet2_hip.point_project(Wf, features)               # [B, C1, N] on csrc/conv1x1.hip
            else:
                P = torch.matmul(Wf, features)
        bn.num_batches_tracked.add_(1)
        return pointnet2_hip.ball_project(P, offsets, W[:, :3], idx, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                          bn.momentum, bn.eps, hasattr(self, 'activation'))

    def forward_ballmax(self, x):
        """relu(bn(conv(x))).max(-1) for x [B, C, npoint, nsample] as ONE autograd node (torch.ops.hcmoco.conv_bn_relu_ballmax,
        csrc/bnact.hip): same values and the same first-maximum rule as this layer followed by F.max_pool2d."""
        from .... import _lib
        bn = self.bn.bn
        bn.num_batches_tracked.add_(1)
        return _lib.torch_glue().conv_bn_relu_ballmax(x, self.conv.weight, bn.weight, bn.bias, bn.running_mean,
                                                      bn.running_var, bn.momentum, bn.eps)


def first_layer_fusable(mlp, xyz):
    """The first layer of ``mlp`` can run on the implicit grouped tensor (``Conv2d.forward_grouped``): training step on the
    MI355X, fp32, conv -> BatchNorm2d [-> ReLU] form (the caller checks the ball shape)."""
    if not (FUSED and len(mlp) > 0 and xyz.is_cuda and xyz.dtype == torch.float32):
        return False
    first = list(mlp)[0]
    return isinstance(first, _ConvBNAct) and first._fusable and first.training


def ballmax_fusable(mlp, x):
    """The last layer of ``mlp`` + the max over the ball can run as one node: training step on the MI355X, fp32, the
    post-activation conv -> BatchNorm2d -> ReLU form, a ball size the kernel has an instance for."""
    if not (FUSED and len(mlp) > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    last = list(mlp)[-1]
    return (isinstance(last, _ConvBNAct) and last._fusable and last.training and hasattr(last, 'activation')
            and x.shape[3] in (4, 8, 16, 32, 64) and x.shape[2] % 4 == 0)


class Conv1d(_ConvBNAct):
    def __init__(self, in_size, out_size, *, bn=False, activation=nn.ReLU(inplace=True)):
        super().__init__(nn.Conv1d, nn.BatchNorm1d, in_size, out_size, bn, activation)


class Conv2d(_ConvBNAct):
    def __init__(self, in_size, out_size, *, bn=False, activation=nn.ReLU(inplace=True)):
        super().__init__(nn.Conv2d, nn.BatchNorm2d, in_size, out_size, bn, activation)


class SharedMLP(nn.Sequential):
    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True)):
        super().__init__()
        for i in range(len(args) - 1):
            self.add_module('layer{}'.format(i), Conv2d(args[i], args[i + 1], bn=bn, activation=activation))
