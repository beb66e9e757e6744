"""SURVEY / VERDICT row a18: the PointNet++ MODULE layer (SA-MSG, SharedMLP, FP modules, ``Pointnet2MSG``) and the HRNetPN
encoder (``depth2pts`` / ``pts2depth`` / forward) against fixtures recorded from the REFERENCE's own Python modules
(tests/golden/gen_golden.py: gen_pointnet2_msg, gen_model_pn_fwd, gen_model_pn_bwd -- the reference's modules unmodified,
their ``pointnet2_cuda`` bound to the C restatement of the reference's kernels, FMA contract).

This file: the helpers, and the product's host-side modules on CPU over the same oracle shim (so the Python layer is
pinned on its own).  tests/test_pn_reference_gpu.py: the DEFAULT GPU runtime (hcm_ball_project_*, hcm_conv1x1_*,
hcm_bn_relu_ballmax_*, the geometry stream, weight-gradient stream on and off) against the same fixtures.

Reference: networks/pointnet2/pointnet2_modules.py:19-55, pytorch_utils.py:5-33, networks/pointnet2_msg.py:79-95,
networks/build_backbone.py:379-455, :457-514."""
import contextlib

import pytest
import torch

from hcmoco_amd.pycontrast.networks.build_backbone import build_model
from hcmoco_amd.pycontrast.networks.pointnet2 import pointnet2_utils
from hcmoco_amd.pycontrast.networks.pointnet2_msg import Pointnet2MSG
from test_model_surface import deterministic_fill, compare_param_grads
from test_hrnetpn import pn_opt

# forward gate: 1e-4 of the element + 1e-5 of the tensor's largest magnitude
FWD_REL, FWD_OF_MAX = 1e-4, 1e-5


def _load_pn_inputs():
    import importlib.util
    import os
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location('pn_inputs', os.path.join(GOLDEN, 'pn_inputs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.pn_inputs


@contextlib.contextmanager
def native(module):
    old = pointnet2_utils.pointnet2
    pointnet2_utils.pointnet2 = module
    try:
        yield
    finally:
        pointnet2_utils.pointnet2 = old


@contextlib.contextmanager
def replay_draw(ind_full):
    """``torch.multinomial`` inside the product's depth2pts returns the pixel draw the reference made."""
    orig = torch.multinomial
    calls = []

    def fake(p, n, replacement=False, **kw):
        assert tuple(p.shape) == (ind_full.shape[0], p.shape[1]) and n == ind_full.shape[1] and replacement
        calls.append(1)
        return ind_full.to(p.device)
    torch.multinomial = fake
    try:
        yield calls
    finally:
        torch.multinomial = orig


def full_draw(ind_kept, mask):
    """The reference draws for the images that have depth only (build_backbone.py:421-427); rows of the others are
    never read (their cloud is zeroed)."""
    keep = mask.reshape(mask.shape[0], -1).sum(-1) > 0
    out = torch.zeros(mask.shape[0], ind_kept.shape[1], dtype=torch.long)
    out[keep] = ind_kept.long()
    return out


def close(got, ref, what, rel=FWD_REL, of_max=FWD_OF_MAX, report=None):
    got, ref = got.detach().float().cpu(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    bound = rel * ref.abs() + of_max * float(ref.abs().max())
    worst = float((err / bound.clamp_min(1e-30)).max())
    if report is not None:
        report[what] = round(worst, 4)
    assert worst <= 1.0, (what, 'error / bound', worst, 'max err', float(err.max()), 'max |ref|', float(ref.abs().max()))


def scalar_close(got, ref, what, rel=FWD_REL):
    assert abs(float(got) - float(ref)) <= rel * abs(float(ref)), (what, float(got), float(ref))


def msg_forward_checks(g, net, device, mode, report=None, levels_bit_exact=True):
    """Per-level centres (FPS picks chained through four levels: bit-exact), per-level features and the output."""
    levels = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: levels.append(out)) for m in net.SA_modules]
    try:
        getattr(net, mode)()
        with torch.no_grad():
            out = net(g['cloud'].to(device))
    finally:
        for h in hooks:
            h.remove()
    for k, (xyz_k, feat_k) in enumerate(levels):
        ref_xyz = g['%s_l%d_xyz' % (mode, k + 1)]
        got_xyz = (xyz_k if k else xyz_k[:, ::16]).cpu()
        assert torch.equal(got_xyz, ref_xyz), ('centres of level', k + 1)
        close(feat_k[:, ::4, ::max(1, feat_k.shape[2] // 64)], g['%s_l%d_feat_slice' % (mode, k + 1)], 'l%d' % (k + 1), report=report)
        scalar_close(feat_k.double().norm(), g['%s_l%d_feat_norm' % (mode, k + 1)], 'l%d norm' % (k + 1))
    # the output sits behind 16 train-mode BatchNorm layers (8 of them in the FP chain, the coarsest over 2 x 64 values): the
    # REFERENCE's own fp32 run is 0.78 of the 1e-5-of-max gate away from the same network in float64 there (0.005 at level 1,
    # tools/probes/pn_truth.py -> profiles/r06_pn_truth.txt), so two fp32 runs can be 1.6 apart: 2e-5 of max for this one tensor in
    # train mode (measured on the MI355X: 0.99 of 1e-5, i.e. 0.5 of this bound), 1e-5 everywhere else
    close(out[:, ::2, ::32], g[mode + '_out_slice'], 'out', of_max=2e-5 if mode == 'train' else FWD_OF_MAX, report=report)
    scalar_close(out.double().norm(), g[mode + '_out_norm'], 'out norm')
    gp = torch.Generator().manual_seed(31)
    proj = torch.randn(out.shape, generator=gp).double()
    dot = float((out.double().cpu() * proj).sum())
    assert abs(dot - float(g[mode + '_out_dot'])) <= FWD_REL * float(g[mode + '_out_norm']) * float(proj.norm()) / out.numel() ** 0.5 * 10, (
        'out projection', dot, float(g[mode + '_out_dot']))
    return out


def msg_backward_checks(g, net, device, tol_full, tol_proj, report=None):
    net.load_state_dict(deterministic_fill(net.state_dict()))
    net.train()
    out = net(g['cloud'].to(device))
    gc = torch.Generator().manual_seed(33)
    cot = torch.randn(out.shape, generator=gc) * 0.1
    loss = (out * cot.to(device)).sum()
    loss.backward()
    if device.type == 'cuda':
        torch.cuda.synchronize()
    mass = float((out.detach() * cot.to(device)).abs().sum())
    loss = loss.detach()
    assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss'])) + 1e-6 * mass, (float(loss), float(g['loss']))
    worst = compare_param_grads(g, net.named_parameters(), tol_full, tol_proj, report)
    sd = net.state_dict()
    # running statistics of the first fused layer after ONE training forward from the filled state (momentum 0.1, unbiased var)
    close(sd['SA_modules.0.mlps.0.layer0.bn.bn.running_mean'], g['bn_running_mean_after'], 'running_mean')
    close(sd['SA_modules.0.mlps.0.layer0.bn.bn.running_var'], g['bn_running_var_after'], 'running_var')
    return worst


def pn_model(device):
    model, _ = build_model(pn_opt())
    model.load_state_dict(deterministic_fill(model.state_dict()))
    return model.to(device)


def pn_forward_checks(g, model, device, report=None):
    d = device
    x, s, mask, grid_xy, mean = (g[k].to(d) for k in ('x', 's', 'depth_mask', 'grid_xy', 'mean'))
    oh, ow = int(g['original_h']), int(g['original_w'])
    ind = full_draw(g['ind'], g['depth_mask'])
    x2 = x[:, 3:]
    with replay_draw(ind):
        sample, full, _ = model.depth2pts(x2, mask, grid_xy, oh, ow, mean)
    # the back-projection is three IEEE products / sums per coordinate in the reference's order: bit-exact
    assert torch.equal(sample[:, :, ::8].cpu(), g['cloud_sample_slice']) and torch.equal(full[:, :, ::8].cpu(), g['cloud_full_slice'])
    assert float(sample.double().sum()) == float(g['cloud_sample_sum']) and float(full.double().sum()) == float(g['cloud_full_sum'])
    assert float(sample[2].abs().sum()) == 0.0            # the image without depth keeps an all-zero cloud
    for mode in ('eval', 'train'):
        getattr(model, mode)()
        with torch.no_grad(), replay_draw(ind) as calls:
            f1, f2, f3, f, aux = model(x, s, mask, grid_xy, oh, ow, mean, return_fm=True)
        if d.type == 'cuda':
            torch.cuda.synchronize()
        assert len(calls) == 1
        tag = mode + ':'
        close(f, g[mode + '_f'], tag + 'f', report=report)
        close(f3, g[mode + '_feat3'], tag + 'feat3', report=report)
        close(f2[:, ::2, ::32], g[mode + '_feat2_slice'], tag + 'feat2', report=report)
        scalar_close(f2.double().norm(), g[mode + '_feat2_norm'], tag + 'feat2 norm')
        close(aux['linear_merge2'][:, ::2], g[mode + '_lm2'], tag + 'linear_merge2', report=report)
        # HRNet side (pinned on its own by test_model_surface*.py): train mode of this 64 x 64 fixture normalises the
        # coarsest branch over 3 x 2 x 2 values, hence 5e-5 of the largest magnitude there (as in test_model_surface_gpu.py)
        hr = 5e-5 if mode == 'train' else 1e-5
        close(aux['linear_merge1'][:, :8, ::5, ::5], g[mode + '_lm1_slice'], tag + 'linear_merge1', of_max=hr, report=report)
        close(f1[3], g[mode + '_feat1_3'], tag + 'feat1[3]', of_max=hr, report=report)
        assert aux['merge2'] is f2


def pn_backward_checks(gb, model, device, tol_full, tol_proj, report=None):
    pn_inputs = _load_pn_inputs()
    x, s_ref, mask, grid_xy, oh, ow, mean = pn_inputs(4, 128, 16, 47, empty=(1,))
    assert torch.equal(s_ref, gb['s']) and abs(float(x.double().sum()) - float(gb['x_checksum'])) < 1e-9
    assert torch.equal(mean, gb['mean'])
    d = device
    model.train()
    s = s_ref.to(d).requires_grad_(True)
    ind = full_draw(gb['ind'], mask)
    with replay_draw(ind):
        f1, f2, f3, f, aux = model(x.to(d), s, mask.to(d), grid_xy.to(d), oh, ow, mean.to(d), return_fm=True)
    gc = torch.Generator().manual_seed(53)
    cf = torch.randn(f.shape, generator=gc)
    c3 = torch.randn(f3.shape, generator=gc) * 0.1
    c1 = torch.randn(aux['linear_merge1'].shape, generator=gc) * 0.05
    c2 = torch.randn(aux['linear_merge2'].shape, generator=gc) * 0.05
    assert torch.equal(cf, gb['cf']) and torch.equal(c3, gb['c3'])
    terms = [f * cf.to(d), f3 * c3.to(d), aux['linear_merge1'] * c1.to(d), aux['linear_merge2'] * c2.to(d)]
    loss = terms[0].sum() + terms[1].sum() + terms[2].sum() + terms[3].sum()
    loss.backward()
    if d.type == 'cuda':
        torch.cuda.synchronize()
    loss = loss.detach()
    close(aux['linear_merge2'][:, ::4, ::3, ::3], gb['lm2_slice'], 'linear_merge2', report=report)
    close(f, gb['f'], 'f', report=report)
    mass = float(gb['mass'])
    assert abs(float(loss) - float(gb['loss'])) <= 1e-4 * abs(float(gb['loss'])) + 1e-6 * mass, (float(loss), float(gb['loss']), mass)
    e = float((s.grad.double().cpu() - gb['grad_s'].double()).norm() / gb['grad_s'].double().norm())
    assert e < tol_full, ('grad_s', e)
    return compare_param_grads(gb, model.named_parameters(), tol_full, tol_proj, report)


# ------------------------------------------------------------------------------------------------------------------ #
# CPU: the product's host-side modules over the oracle shim
# ------------------------------------------------------------------------------------------------------------------ #
def test_state_dict_keys_of_the_cloud_encoder(golden):
    g = golden('pointnet2_msg')
    assert list(Pointnet2MSG(input_channels=0).state_dict().keys()) == [str(k) for k in g['keys']]


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_host_modules_forward_match_reference(golden, mode):
    from oracle import pointnet2_shim
    g = golden('pointnet2_msg')
    net = Pointnet2MSG(input_channels=0)
    net.load_state_dict(deterministic_fill(net.state_dict()))
    report = {}
    with native(pointnet2_shim):
        msg_forward_checks(g, net, torch.device('cpu'), mode, report)
    print(mode, 'error / bound:', report)


def test_host_modules_backward_match_reference(golden):
    from oracle import pointnet2_shim
    g = golden('pointnet2_msg')
    net = Pointnet2MSG(input_channels=0)
    report = {}
    with native(pointnet2_shim):
        worst = msg_backward_checks(g, net, torch.device('cpu'), tol_full=1e-6, tol_proj=1e-5, report=report)   # measured: exactly 0 (same ATen ops, same C kernels)
    print(worst)


def test_hrnetpn_host_forward_matches_reference(golden):
    from oracle import pointnet2_shim
    report = {}
    with native(pointnet2_shim):
        pn_forward_checks(golden('model_hrnetpn_w18_mpii'), pn_model(torch.device('cpu')), torch.device('cpu'), report)
    print('error / bound:', report)


def test_hrnetpn_host_backward_matches_reference(golden):
    from oracle import pointnet2_shim
    report = {}
    with native(pointnet2_shim):
        worst = pn_backward_checks(golden('model_bwd_hrnetpn_w18_mpii'), pn_model(torch.device('cpu')), torch.device('cpu'),
                                   tol_full=1e-6, tol_proj=1e-5, report=report)   # measured: exactly 0
    print(worst)
