"""hcm_conv3x3_forward / hcm_conv3x3_backward_data (direct fp32 MFMA convolution of the two high-resolution
HRNet branches) against torch's conv2d in float64 -- what networks/official_hrnet.py:40-70 `conv3x3` computes.
A floating-point kernel: tolerance 2e-5 of the output's scale (fp32 sums of 162 / 324 products)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _call(name, a, w, N, Cc, K, H, W):
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    out = torch.full((N, K if name.endswith('forward') else Cc, H, W), float('nan'), device=a.device)
    check(getattr(_lib.lib(), name)(C.c_void_p(a.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()),
                                    N, Cc, K, H, W, C.c_void_p(torch.cuda.current_stream().cuda_stream)), name)
    return out


@pytest.mark.parametrize('N,Cc,H,W', [(32, 18, 64, 64), (32, 36, 32, 32), (3, 18, 8, 64), (2, 20, 12, 64), (1, 17, 4, 64),
                                      (5, 36, 4, 32), (2, 33, 20, 32), (1, 34, 32, 32)])
def test_conv3x3_forward_and_data_gradient_match_conv2d(N, Cc, H, W):
    from hcmoco_amd import _lib
    assert _lib.lib().hcm_conv3x3_supported(Cc, Cc, H, W) == 1
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N * 1000 + Cc)
    x = torch.randn(N, Cc, H, W, generator=g).to(dev)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.1).to(dev)
    gy = torch.randn(N, Cc, H, W, generator=g).to(dev)
    y = _call('hcm_conv3x3_forward', x, w, N, Cc, Cc, H, W)
    dx = _call('hcm_conv3x3_backward_data', gy, w, N, Cc, Cc, H, W)
    xd = x.double().requires_grad_()
    yd = F.conv2d(xd, w.double(), padding=1)
    yd.backward(gy.double())
    for got, ref in ((y, yd.detach()), (dx, xd.grad)):
        assert torch.isfinite(got).all()
        scale = ref.abs().max().item()
        assert (got.double() - ref).abs().max().item() <= 2e-5 * scale


def test_conv3x3_is_deterministic_and_refuses_other_shapes():
    from hcmoco_amd import _lib
    dev = torch.device('cuda:0')
    x, w = torch.randn(4, 18, 64, 64, device=dev), torch.randn(18, 18, 3, 3, device=dev)
    a = _call('hcm_conv3x3_forward', x, w, 4, 18, 18, 64, 64)
    b = _call('hcm_conv3x3_forward', x, w, 4, 18, 18, 64, 64)
    assert torch.equal(a, b)
    L = _lib.lib()
    for shape in [(72, 72, 16, 16), (18, 36, 64, 64), (18, 18, 62, 64), (18, 18, 64, 48), (16, 16, 64, 64), (32, 32, 64, 64)]:
        assert L.hcm_conv3x3_supported(*shape) == 0
    with pytest.raises(_lib.HipError):
        _call('hcm_conv3x3_forward', torch.randn(1, 72, 16, 16, device=dev), torch.randn(72, 72, 3, 3, device=dev),
              1, 72, 72, 16, 16)


@pytest.mark.parametrize('N,Cc,H,W', [(32, 18, 64, 64), (32, 36, 32, 32), (3, 17, 8, 64), (2, 20, 12, 64), (5, 34, 16, 32)])
def test_conv3x3_forward_stats_and_the_batch_norm_that_consumes_them(N, Cc, H, W):
    """hcm_conv3x3_forward_stats: same y as hcm_conv3x3_forward bit for bit, and per-slot sums of y / y^2 that add up
    to the tensor's own (float64) sums; hcm_bn_act_forward_pre on those partials == hcm_bn_act_forward on y (2e-5)."""
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N + Cc)
    x = (torch.randn(N, Cc, H, W, generator=g) + 0.5).to(dev)
    w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.1).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    y0 = _call('hcm_conv3x3_forward', x, w, N, Cc, Cc, H, W)
    slots = int(L.hcm_conv3x3_stats_slots(N, H))
    assert slots == N * H // 4
    y = torch.empty_like(y0)
    part = torch.full((2 * slots + 1, Cc), float('nan'), device=dev)
    check(L.hcm_conv3x3_forward_stats(p(x), p(w), p(y), N, Cc, Cc, H, W, None, p(part), st), 'forward_stats')
    assert torch.equal(y, y0)
    assert torch.equal(part[-1], torch.zeros(Cc, device=dev))          # the shift that was used: none
    s1 = part[:-1][0::2].double().sum(0)
    s2 = part[:-1][1::2].double().sum(0)
    yd = y.double()
    r1, r2 = yd.sum((0, 2, 3)), yd.square().sum((0, 2, 3))
    assert (s1 - r1).abs().max().item() <= 1e-5 * r2.sqrt().max().item() * (N * H * W) ** 0.5
    assert ((s2 - r2).abs() / r2).max().item() <= 1e-5
    if N * H * W <= 8192:                                    # the one-workgroup BN form has no separate statistics pass
        return
    gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), torch.randn(Cc, generator=g).to(dev)
    res = torch.randn(N, Cc, H, W, generator=g).to(dev)
    outs = []
    for pre in (False, True, 'shifted'):
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        out = torch.empty_like(y)
        stats = torch.empty(int(L.hcm_bn_act_stats_floats(N, Cc, H * W)), device=dev)
        if pre == 'shifted':
            shift = (y.mean((0, 2, 3)) * 0.9 + 0.05).contiguous()
            part2 = torch.full((2 * slots + 1, Cc), float('nan'), device=dev)
            check(L.hcm_conv3x3_forward_stats(p(x), p(w), p(y), N, Cc, Cc, H, W, p(shift), p(part2), st), 'forward_stats')
            assert torch.equal(y, y0) and torch.equal(part2[-1], shift)
            check(L.hcm_bn_act_forward_pre(p(y), p(res), p(gamma), p(beta), p(rm), p(rv), 0.1, 1e-5, 1, N, Cc, H * W, p(out),
                                           p(stats), p(part2), slots, st), 'bn_pre')
        elif pre:
            check(L.hcm_bn_act_forward_pre(p(y), p(res), p(gamma), p(beta), p(rm), p(rv), 0.1, 1e-5, 1, N, Cc, H * W, p(out),
                                           p(stats), p(part), slots, st), 'bn_pre')
        else:
            check(L.hcm_bn_act_forward(p(y), p(res), p(gamma), p(beta), p(rm), p(rv), 0.1, 1e-5, 1, N, Cc, H * W, p(out),
                                       p(stats), st), 'bn')
        outs.append((out, stats[:2 * Cc].clone(), rm, rv))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            scale = a.abs().max().item()
            assert (a - b).abs().max().item() <= 2e-5 * scale


def test_batch_norm_statistics_from_the_conv_epilogue_survive_a_large_mean():
    """|mean| >> std (E[y^2] - E[y]^2 from unshifted fp32 sums cancels): a centre-tap filter on inputs
    100 + 0.01 noise gives y ~ 90 +- 0.0025 per channel.  With the sums taken about a running mean within 0.05 of the
    batch mean, hcm_bn_act_forward_pre's mean / invstd agree with float64 statistics of y to 1e-3; the two-pass
    hcm_bn_act_forward (shifted by the first element) is the yardstick."""
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    L = _lib.lib()
    dev = torch.device('cuda:0')
    N, Cc, H, W = 32, 18, 64, 64
    g = torch.Generator().manual_seed(5)
    x = (100 + 0.01 * torch.randn(N, Cc, H, W, generator=g)).to(dev)
    w = torch.zeros(Cc, Cc, 3, 3)
    w[:, :, 1, 1] = torch.rand(Cc, Cc, generator=g) * 0.1
    w = w.to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    slots = int(L.hcm_conv3x3_stats_slots(N, H))
    y = torch.empty(N, Cc, H, W, device=dev)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    got = {}
    for name, shift in (('plain', None), ('shifted', 'near')):
        part = torch.empty(2 * slots + 1, Cc, device=dev)
        if shift is not None:
            check(L.hcm_conv3x3_forward(p(x), p(w), p(y), N, Cc, Cc, H, W, st), 'conv')
            shift = (y.mean((0, 2, 3)) + 0.05).contiguous()
        check(L.hcm_conv3x3_forward_stats(p(x), p(w), p(y), N, Cc, Cc, H, W, p(shift), p(part), st), 'forward_stats')
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        out = torch.empty_like(y)
        stats = torch.empty(int(L.hcm_bn_act_stats_floats(N, Cc, H * W)), device=dev)
        check(L.hcm_bn_act_forward_pre(p(y), None, p(gamma), p(beta), p(rm), p(rv), 0.1, 1e-5, 0, N, Cc, H * W, p(out), p(stats),
                                       p(part), slots, st), 'bn_pre')
        got[name] = stats[:2 * Cc].clone()
    yd = y.double()
    mean = yd.mean((0, 2, 3))
    invstd = 1.0 / (yd.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()
    assert float(mean.abs().min()) > 30 and float(invstd.min()) > 100          # the regime this test is about
    sh = got['shifted'].double()
    assert float(((sh[:Cc] - mean) / mean).abs().max()) < 1e-6
    assert float((sh[Cc:] / invstd - 1).abs().max()) < 1e-3
    print('plain sums: invstd off by up to %.3g (relative)' % float((got['plain'].double()[Cc:] / invstd - 1).abs().max()))
