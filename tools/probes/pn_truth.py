"""Diagnostic (r06, row a18): how far is the fp32 REFERENCE fixture of the cloud encoder from exact arithmetic?

Runs the product's host-side Pointnet2MSG on CPU with the feature path in float64 (geometry -- FPS, ball query, three_nn
-- stays the fp32 C restatement, so the indices are the fixture's) and compares the fixture's fp32 train-mode slices with
it in units of the forward gate (1e-4 of the element + 1e-5 of the tensor's largest magnitude).  Train-mode BatchNorm
divides every layer's round-off by the batch standard deviation, so two fp32 implementations that sum in different
orders drift apart along the 8-layer FP chain; this prints what the reference's own fp32 run carries.
CPU only; uses oracle/ (diagnostic tooling, never the product path)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from hcmoco_amd.pycontrast.networks.pointnet2 import pointnet2_utils as U   # noqa: E402
from hcmoco_amd.pycontrast.networks.pointnet2_msg import Pointnet2MSG       # noqa: E402
from oracle import pointnet2_shim                                            # noqa: E402
from test_model_surface import deterministic_fill                            # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pointnet2_msg.npz'))
    U.pointnet2 = pointnet2_shim
    f32 = {k: getattr(U, k) for k in ('furthest_point_sample', 'ball_query', 'three_nn', 'gather_operation',
                                      'grouping_operation', 'three_interpolate')}
    U.furthest_point_sample = lambda xyz, n: f32['furthest_point_sample'](xyz.float(), n)
    U.ball_query = lambda r, ns, xyz, new: f32['ball_query'](r, ns, xyz.float().contiguous(), new.float().contiguous())

    def three_nn(unknown, known):
        d, i = f32['three_nn'](unknown.float().contiguous(), known.float().contiguous())
        return d.double(), i
    U.three_nn = three_nn
    U.gather_operation = lambda feats, idx: torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, feats.shape[1], -1))

    def group(feats, idx):
        B, C, N = feats.shape
        _, M, S = idx.shape
        return torch.gather(feats, 2, idx.long().reshape(B, 1, M * S).expand(B, C, M * S)).reshape(B, C, M, S)
    U.grouping_operation = group

    def interp(feats, idx, w):
        B, C, M = feats.shape
        n = idx.shape[1]
        got = torch.gather(feats, 2, idx.long().reshape(B, 1, n * 3).expand(B, C, n * 3)).reshape(B, C, n, 3)
        return (got * w.unsqueeze(1)).sum(-1)
    U.three_interpolate = interp

    net = Pointnet2MSG(input_channels=0)
    net.load_state_dict(deterministic_fill(net.state_dict()))
    net = net.double()
    cloud = torch.from_numpy(g['cloud']).double()
    # the centres must be the fp32 values: xyz enters the features as fp32 numbers cast up
    for mode in ('eval', 'train'):
        getattr(net, mode)()
        levels = []
        hooks = [m.register_forward_hook(lambda mod, inp, out: levels.append(out)) for m in net.SA_modules]
        with torch.no_grad():
            out = net(cloud)
        for h in hooks:
            h.remove()

        def units(ref32, truth):
            ref32 = torch.from_numpy(ref32).double()
            bound = 1e-4 * ref32.abs() + 1e-5 * float(ref32.abs().max())
            return float(((ref32 - truth).abs() / bound).max())
        rep = {}
        for k, (xyz_k, feat_k) in enumerate(levels):
            rep['l%d' % (k + 1)] = round(units(g['%s_l%d_feat_slice' % (mode, k + 1)], feat_k[:, ::4, ::max(1, feat_k.shape[2] // 64)]), 4)
        rep['out'] = round(units(g[mode + '_out_slice'], out[:, ::2, ::32]), 4)
        print(mode, 'reference fp32 fixture vs float64 feature path, error / gate:', rep)
    # backward: the reference's fp32 parameter gradients (fixture, stored in full for the small tensors) vs float64
    net.load_state_dict({k: v.double() for k, v in deterministic_fill(net.float().state_dict()).items()})
    net = net.double().train()
    out = net(cloud)
    cot = (torch.randn(out.shape, generator=torch.Generator().manual_seed(33)) * 0.1).double()
    (out * cot).sum().backward()
    rows = []
    for k, p_ in net.named_parameters():
        if 'g:' + k in g.files:
            ref = torch.from_numpy(g['g:' + k]).double()
            rows.append((float((ref - p_.grad).norm() / p_.grad.norm().clamp_min(1e-30)), k, float(p_.grad.norm())))
    rows.sort(reverse=True)
    print('reference fp32 gradients vs float64, relative L2, worst first:')
    for r in rows[:10]:
        print('  %.3e  %-50s |g| = %.4g' % r)
    torch.save({k: p_.grad.clone() for k, p_ in net.named_parameters() if p_.numel() <= 2048}, '/tmp/pn_truth_grads.pt')


if __name__ == '__main__':
    main()
