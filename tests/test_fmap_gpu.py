"""GPU parity of the feature-map loss kernels through the C ABI: against the
golden vectors generated from the reference, and against the CPU oracle at other sizes/layouts.
Tolerances (fp32): losses 1e-5 relative (+1e-6 abs), accuracies exact, gradients 1e-4 rel-L2."""
import math

import pytest
import torch

from oracle import hcmoco_oracle as O

pytestmark = pytest.mark.gpu
LOSS_RTOL, GRAD_REL_L2 = 1e-5, 1e-4


def d():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def ops():
    from hcmoco_amd import hip_ops
    return hip_ops


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def expand_ind(sample_ind, keep, B):
    """reference sample_ind is [B', S] for kept images; the ABI takes [B, S]."""
    out = torch.zeros(B, sample_ind.shape[1], dtype=torch.int64)
    out[keep] = sample_ind
    return out


def run(map1, map2, feat3, sample_ind, keep, pix, vis, ud, ur, temp, layout, **which):
    dev = d()

    def mv(t):
        t = t.to(dev)
        if layout == 'channels_last':
            t = t.contiguous(memory_format=torch.channels_last)
        return t.requires_grad_(True)
    m1, m2 = mv(map1), mv(map2)
    f3 = None if feat3 is None else feat3.to(dev).requires_grad_(True)
    to = lambda t: None if t is None else t.to(dev)
    total, meters = ops().fmap_losses(m1, m2, f3, to(sample_ind), to(keep), to(pix), to(vis), to(ud), to(ur),
                                      temp, **which)
    total.backward()
    return total, meters.cpu(), m1.grad, m2.grad, (None if f3 is None else f3.grad)


@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_dense_vs_golden(golden, layout):
    g = golden('dense_soft_nce')
    keep, _ = O.dense_keep(g['depth_mask'], g['h'], g['h'])
    ind = expand_ind(g['sample_ind'], keep, g['B'])
    total, met, g1, g2, _ = run(g['map1'], g['map2'], None, ind, keep.int(), None, None, None, None,
                                g['temperature'], layout, do_dense=True, do_joint=False, do_scl=False)
    assert torch.allclose(met[0:2], g['losses'], rtol=LOSS_RTOL, atol=1e-6), (met[:4], g['losses'])
    assert torch.allclose(met[2:4], g['accs'], rtol=0, atol=1e-6)
    assert rel_l2(g1, g['grad_map1']) < GRAD_REL_L2 and rel_l2(g2, g['grad_map2']) < GRAD_REL_L2
    assert float(g1[3].abs().max()) == 0
    assert abs(float(total) - float(g['losses'].sum())) < 1e-4


def test_dense_nothing_kept_gives_zeros(golden):
    g = golden('dense_soft_nce')
    ind = torch.zeros(g['B'], g['S'], dtype=torch.int64)
    total, met, g1, g2, _ = run(g['map1'], g['map2'], None, ind, torch.zeros(g['B'], dtype=torch.int32), None, None,
                                None, None, 0.07, 'nchw', do_dense=True, do_joint=False, do_scl=False)
    assert float(met[:4].abs().max()) == 0 and float(g1.abs().max()) == 0 and float(g2.abs().max()) == 0


@pytest.mark.parametrize('J', [13, 16, 17])
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_joint_vs_golden(golden, J, layout):
    g = golden('joint_nce')
    p = 'J%d_' % J
    pix = O.joint_pixels(g[p + 'joints2d'], 8)
    gpix = ops().joint_pixels(g[p + 'joints2d'].to(d()), 8)
    assert torch.equal(gpix.cpu(), pix)                                   # bit-exact index bookkeeping
    assert torch.equal(ops().joint_pixels(g[p + 'joints2d'].double().to(d()), 8).cpu(), pix)
    total, met, g1, g2, g3 = run(g[p + 'map1'], g[p + 'map2'], g[p + 'feat3'], None, None, gpix.cpu(),
                                 g[p + 'joints_vis'], g[p + 'use_depth'], None, g['temperature'], layout,
                                 do_dense=False, do_joint=True, do_scl=False)
    assert torch.allclose(met[4:6], g[p + 'losses'], rtol=LOSS_RTOL, atol=1e-6)
    assert torch.allclose(met[6:8], g[p + 'accs'], rtol=0, atol=1e-6)
    assert rel_l2(g1, g[p + 'grad_map1']) < GRAD_REL_L2
    assert rel_l2(g2, g[p + 'grad_map2']) < GRAD_REL_L2
    assert rel_l2(g3, g[p + 'grad_feat3']) < GRAD_REL_L2


def test_joint_all_ignored_is_nan_with_zero_grads():
    torch.manual_seed(9)
    m = torch.randn(2, 128, 8, 8)
    j2d = torch.rand(2, 16, 2) * 32
    total, met, g1, g2, g3 = run(m, m.clone(), torch.randn(2, 16, 128), None, None, O.joint_pixels(j2d, 8),
                                 torch.ones(2, 16).int(), torch.zeros(2).int(), None, 0.07, 'nchw',
                                 do_dense=False, do_joint=True, do_scl=False)
    assert not math.isnan(float(met[4])) and math.isnan(float(met[5]))     # reference: CE over no target


@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_scl_vs_golden(golden, layout):
    g = golden('scl')
    pix = O.joint_pixels(g['joints2d'], 8)
    for tag, ur in (('with_rgb', g['use_rgb']), ('rgb_none', None)):
        total, met, g1, g2, _ = run(g['map1'], g['map2'], None, None, None, pix, None, g['use_depth'], ur,
                                    g['temperature'], layout, do_dense=False, do_joint=False, do_scl=True)
        ref = float(g['loss_' + tag])
        assert abs(float(met[8]) - ref) < LOSS_RTOL * abs(ref) + 1e-6
        assert rel_l2(g1, g['grad1_' + tag]) < GRAD_REL_L2 and rel_l2(g2, g['grad2_' + tag]) < GRAD_REL_L2
    total, met, g1, g2, _ = run(g['map1'], g['map2'], None, None, None, pix, None, torch.zeros(4).int(), None,
                                g['temperature'], layout, do_dense=False, do_joint=False, do_scl=True)
    assert float(met[8]) == 0 and float(g1.abs().max()) == 0              # use_depth.sum()==0 early-out


@pytest.mark.parametrize('B,h,S,J', [(2, 8, 1, 1), (3, 12, 17, 5), (4, 16, 100, 17), (32, 64, 400, 17)])
def test_all_three_vs_oracle(B, h, S, J):
    """Ragged / full BASELINE sizes (last case: B=32, 64x64 maps, S=400, J=17), all losses into one
    pair of gradient buffers, duplicates among sampled pixels and joints."""
    torch.manual_seed(B * 100 + S)
    C, temp = 128, 0.07
    m1, m2 = torch.randn(B, C, h, h), torch.randn(B, C, h, h)
    f3 = torch.randn(B, J, C)
    keep = torch.ones(B, dtype=torch.bool)
    if B > 2:
        keep[1] = False
    ud = keep.clone().int()
    ind = torch.randint(0, h * h, (B, S))
    ind[0, S // 2] = ind[0, 0]                                   # duplicates inside an image
    j2d = torch.rand(B, J, 2) * 4 * h * 1.1 - 2
    if J > 1:
        j2d[0, 1] = j2d[0, 0]
    vis = (torch.rand(B, J) < 0.85).int()
    vis[0, 0] = 1
    pix = O.joint_pixels(j2d, h)
    total, met, g1, g2, g3 = run(m1, m2, f3, ind, keep.int(), pix, vis, ud, None, temp, 'channels_last')
    ld, ad, d1, d2 = O.dense_soft_nce(m1, m2, ind[keep], keep, temp, ud)
    lj, aj, j1, j2, j3 = O.joint_nce(m1, m2, f3, j2d, vis, temp, ud)
    ls, s1, s2, _ = O.scl(m1, m2, j2d, temp, ud, None)
    assert torch.allclose(met[0:2], ld, rtol=LOSS_RTOL, atol=1e-6)
    assert torch.allclose(met[2:4], ad, rtol=0, atol=1e-6)
    assert torch.allclose(met[4:6], lj, rtol=LOSS_RTOL, atol=1e-6)
    assert torch.allclose(met[6:8], aj, rtol=0, atol=1e-6)
    assert abs(float(met[8]) - float(ls)) < LOSS_RTOL * abs(float(ls)) + 1e-6
    assert rel_l2(g1, d1 + j1 + s1) < GRAD_REL_L2
    assert rel_l2(g2, d2 + j2 + s2) < GRAD_REL_L2
    assert rel_l2(g3, j3) < GRAD_REL_L2


def test_exact_fp32_mode_is_selectable_and_closer_to_float64():
    """``gemm_dtype='fp32_exact'`` (r06; ADVICE r05: the default fp32 path is a split-bf16 contraction, a true-fp32 one must stay
    selectable): three bf16 pieces per operand, all nine piece products.  Against the float64 oracle at the BASELINE shape
    (B = 32, 64 x 64 maps, S = 400, J = 17) it must be at least as close as the default on every loss and gradient, and the
    default must stay inside the SURVEY 8d gate.  Prints both."""
    torch.manual_seed(77)
    B, h, S, J, C, temp = 32, 64, 400, 17, 128, 0.07
    m1, m2 = torch.randn(B, C, h, h), torch.randn(B, C, h, h)
    f3 = torch.randn(B, J, C)
    keep = torch.ones(B, dtype=torch.bool)
    keep[1] = False
    ud = keep.clone().int()
    ind = torch.randint(0, h * h, (B, S))
    j2d = torch.rand(B, J, 2) * 4 * h
    vis = (torch.rand(B, J) < 0.85).int()
    vis[0, 0] = 1
    pix = O.joint_pixels(j2d, h)
    ld, ad, d1, d2 = O.dense_soft_nce(m1.double(), m2.double(), ind[keep], keep, temp, ud)
    ls, s1, s2, _ = O.scl(m1.double(), m2.double(), j2d, temp, ud, None)
    want_l = torch.cat([ld.double(), ls.double().reshape(1)])
    err = {}
    for mode in ('fp32', 'fp32_exact'):
        total, met, g1, g2, _ = run(m1, m2, f3, ind, keep.int(), pix, vis, ud, None, temp, 'nchw', do_joint=False,
                                    gemm_dtype=mode)
        got_l = torch.cat([met[0:2].double(), met[8:9].double()])
        err[mode] = (float(((got_l - want_l).abs() / want_l.abs()).max()), rel_l2(g1, d1 + s1), rel_l2(g2, d2 + s2))
    print('loss rel / grad1 rel-L2 / grad2 rel-L2:', err)
    assert err['fp32'][0] < LOSS_RTOL and max(err['fp32'][1:]) < GRAD_REL_L2
    assert err['fp32_exact'][0] <= max(err['fp32'][0], 2e-7)
    assert max(err['fp32_exact'][1:]) <= max(err['fp32'][1:]) and max(err['fp32_exact'][1:]) < 2e-6


def test_determinism_bitwise():
    """Owner-computes scatter and fixed-order reductions: two runs are bit-identical."""
    torch.manual_seed(5)
    B, h, S, J = 8, 16, 200, 17
    m1, m2, f3 = torch.randn(B, 128, h, h), torch.randn(B, 128, h, h), torch.randn(B, J, 128)
    ind = torch.randint(0, 40, (B, S))                            # many duplicate pixels
    j2d = torch.rand(B, J, 2) * 8
    pix = O.joint_pixels(j2d, h)
    args = (m1, m2, f3, ind, torch.ones(B).int(), pix, torch.ones(B, J).int(), torch.ones(B).int(), None, 0.07, 'nchw')
    a = run(*args)
    b = run(*args)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


@pytest.mark.parametrize('shape,size', [((2, 5, 8, 8), (16, 16)), ((3, 7, 4, 4), (32, 32)), ((1, 3, 5, 7), (11, 13)),
                                        ((2, 4, 16, 16), (16, 16)), ((32, 36, 32, 32), (64, 64)), ((2, 2, 9, 9), (4, 6))])
def test_upsample_bilinear_matches_aten(shape, size):
    """Row 8 helper vs the plain PyTorch fp32 op, forward and backward."""
    torch.manual_seed(1)
    x = torch.randn(*shape, device=d(), requires_grad=True)
    y = ops().upsample_bilinear(x, size)
    ref = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=False)
    assert y.shape == ref.shape
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)
    g = torch.randn_like(ref)
    gx, = torch.autograd.grad(y, x, g)
    gr, = torch.autograd.grad(ref, x, g)
    assert torch.allclose(gx, gr, rtol=1e-4, atol=1e-5)
    # channels-last input stays channels-last
    xc = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yc = ops().upsample_bilinear(xc, size)
    assert torch.allclose(yc, ref, rtol=1e-5, atol=1e-6)
    if shape[1] > 1 and size[0] * size[1] > 1:
        assert yc.is_contiguous(memory_format=torch.channels_last)
    gc, = torch.autograd.grad(yc, xc, g)
    assert torch.allclose(gc, gr, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
def test_sampled_projection_equals_full_projection(layout):
    """Row 8 sampled form: conv1x1(merge_all_res(branches))[pixels] == sampled_projection(branches),
    values and gradients wrt every branch map, the conv weight and bias (vs plain PyTorch ops)."""
    torch.manual_seed(7)
    dev = d()
    B, h, R = 3, 16, 37
    chans = [18, 36, 72, 144]
    conv = torch.nn.Conv2d(sum(chans), 128, 1).to(dev)
    maps = []
    for i, c in enumerate(chans):
        m = torch.randn(B, c, h >> i, h >> i, device=dev)
        if layout == 'channels_last':
            m = m.contiguous(memory_format=torch.channels_last)
        maps.append(m.requires_grad_(True))
    pix = torch.randint(0, h * h, (B, R), device=dev)
    pix[0, 1] = pix[0, 0]
    # reference data flow in stock PyTorch
    up = [maps[0]] + [torch.nn.functional.interpolate(m, size=(h, h), mode='bilinear', align_corners=False)
                      for m in maps[1:]]
    full = conv(torch.cat(up, 1))                                        # [B,128,h,h]
    ref = torch.gather(full.reshape(B, 128, h * h), 2, pix.unsqueeze(1).expand(B, 128, R)).permute(0, 2, 1)
    g = torch.randn_like(ref)
    ref_grads = torch.autograd.grad(ref, maps + [conv.weight, conv.bias], g)
    rows = ops().sampled_projection(conv.weight, conv.bias, pix, maps)
    assert torch.allclose(rows, ref, rtol=1e-4, atol=1e-5)
    grads = torch.autograd.grad(rows, maps + [conv.weight, conv.bias], g)
    for got, want in zip(grads, ref_grads):
        assert rel_l2(got, want) < 1e-4


def test_losses_on_sampled_rows_equal_losses_on_full_maps():
    """engine.fmap_sampled (branches -> sampled rows -> losses) == engine.fmap (full maps)."""
    from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
    torch.manual_seed(11)
    dev = d()
    B, h, S, J = 4, 16, 40, 17
    chans = [18, 36, 72, 144]
    convs = [torch.nn.Conv2d(sum(chans), 128, 1).to(dev) for _ in range(2)]
    br = [[torch.randn(B, c, h >> i, h >> i, device=dev, requires_grad=True) for i, c in enumerate(chans)]
          for _ in range(2)]
    feat3 = torch.randn(B, J, 128, device=dev, requires_grad=True)
    mask = torch.zeros(B, 4 * h, 4 * h, device=dev)
    mask[:, 8:40, 8:40] = 1
    mask[2] = 0
    j2d = torch.rand(B, J, 2, device=dev) * 4 * h
    vis = (torch.rand(B, J, device=dev) < 0.85).int()
    ud = torch.tensor([1, 1, 0, 1], device=dev)
    eng = HipLossEngine()
    ind, keep = eng.dense_samples(mask, h, h, S, ud)
    leaves = br[0] + br[1] + [feat3] + [p for c in convs for p in c.parameters()]

    def full():
        up = lambda ms: torch.cat([ms[0]] + [torch.nn.functional.interpolate(m, size=(h, h), mode='bilinear',
                                                                            align_corners=False) for m in ms[1:]], 1)
        return eng.fmap(convs[0](up(br[0])), convs[1](up(br[1])), feat3, mask, j2d, vis, ud, None, S, 0.07,
                        sample_ind=ind, keep=keep)
    t1, m1 = full()
    g1 = torch.autograd.grad(t1, leaves)
    t2, m2 = eng.fmap_sampled(br[0], br[1], convs[0], convs[1], feat3, mask, j2d, vis, ud, None, S, 0.07,
                              sample_ind=ind, keep=keep)
    g2 = torch.autograd.grad(t2, leaves)
    assert torch.allclose(m1, m2, rtol=2e-5, atol=1e-6), (m1, m2)
    for a, b in zip(g2, g1):
        assert rel_l2(a, b) < 2e-4


@pytest.mark.parametrize('shape,size', [((32, 18, 8, 8), (64, 64)), ((32, 18, 16, 16), (64, 64)), ((32, 36, 8, 8), (32, 32)),
                                        ((4, 18, 32, 32), (64, 64)), ((2, 3, 7, 5), (21, 20)), ((2, 3, 6, 6), (6, 6)),
                                        ((1, 2, 1, 1), (4, 4)), ((2, 2, 12, 10), (5, 3)),
                                        ((1, 2, 128, 128), (256, 256))])     # too large for LDS: per-pixel kernel
def test_upsample_backward_gather_form(shape, size):
    """hcm_upsample_bilinear2d_backward (row 8's backward, HRNet fuse layers): equals the float64 backward of
    F.interpolate to fp32 round-off -- up-sampling by 2/4/8, non-integer factors, identity and down-sampling --
    and is bit-identical run to run (ATen's kernel scatters with float atomics)."""
    torch.manual_seed(3)
    x = torch.randn(*shape, device=d(), requires_grad=True)
    g = torch.randn(shape[0], shape[1], *size, device=d())
    y = ops().upsample_bilinear(x, size)
    gx1, = torch.autograd.grad(y, x, g, retain_graph=True)
    gx2, = torch.autograd.grad(y, x, g)
    assert torch.equal(gx1, gx2)
    x64 = x.detach().double().requires_grad_(True)
    ref = torch.nn.functional.interpolate(x64, size=size, mode='bilinear', align_corners=False)
    gr, = torch.autograd.grad(ref, x64, g.double())
    scale = float(gr.abs().max()) + 1e-12
    assert float((gx1.double() - gr).abs().max()) <= 2e-6 * scale, float((gx1.double() - gr).abs().max()) / scale
    # sum-preserving: every output pixel's weights add up to 1
    assert abs(float(gx1.double().sum()) - float(g.double().sum())) <= 1e-5 * float(g.abs().double().sum())


def test_sampled_loss_section_is_bitwise_deterministic():
    """Rows 5-8 as the trainer runs them (engine.fmap_sampled: finest branch through hcm_sample_rows with its
    owner-computes backward, coarse branches through sampling matrices + library GEMMs, 1x1 projection, the
    three loss kernels with key-split SCL): two passes from the same inputs give bit-identical losses and
    gradients for all eight branch maps, both projections and the graph features."""
    from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
    torch.manual_seed(12)
    dev = d()
    B, h, S, J = 8, 32, 100, 17
    chans = [18, 36, 72, 144]
    convs = [torch.nn.Conv2d(sum(chans), 128, 1).to(dev) for _ in range(2)]
    br = [[torch.randn(B, c, h >> i, h >> i, device=dev, requires_grad=True) for i, c in enumerate(chans)]
          for _ in range(2)]
    feat3 = torch.randn(B, J, 128, device=dev, requires_grad=True)
    mask = torch.zeros(B, 4 * h, 4 * h, device=dev)
    mask[:, 8:24, 8:24] = 1                                     # 16 mask pixels on the map: many duplicate samples
    j2d = torch.rand(B, J, 2, device=dev) * 4 * h
    j2d[:, 1] = j2d[:, 0]                                       # two joints on one pixel
    vis = (torch.rand(B, J, device=dev) < 0.85).int()
    ud = torch.ones(B, dtype=torch.long, device=dev)
    eng = HipLossEngine()
    ind, keep = eng.dense_samples(mask, h, h, S, ud)
    leaves = br[0] + br[1] + [feat3] + [p for c in convs for p in c.parameters()]
    runs = []
    for _ in range(2):
        t, m = eng.fmap_sampled(br[0], br[1], convs[0], convs[1], feat3, mask, j2d, vis, ud, None, S, 0.07,
                                sample_ind=ind, keep=keep)
        runs.append([t.detach().clone(), m.clone()] + [g_.clone() for g_ in torch.autograd.grad(t, leaves)])
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('B,h,S,J', [(3, 12, 17, 5), (4, 16, 100, 17), (32, 64, 400, 17)])
def test_bf16_mfma_contractions_vs_fp32_oracle(B, h, S, J):
    """BASELINE config 5: dense + SCL with both strip-kernel contractions on the bf16 matrix cores
    (gemm_dtype='bf16'), against the fp32 oracle at the restated tolerance -- losses 1e-2 relative, accuracies
    within one sample, gradients 2e-2 relative L2 -- and against the oracle evaluated on correspondingly
    perturbed inputs is NOT needed: the fp32 softmax statistics keep the error at operand-rounding level
    (2^-9 per logit / tau).  Last case = bench size (B=32, 64x64 maps, S=400, J=17)."""
    torch.manual_seed(B * 10 + S)
    C, temp = 128, 0.07
    m1, m2 = torch.randn(B, C, h, h), torch.randn(B, C, h, h)
    keep = torch.ones(B, dtype=torch.bool)
    keep[1] = False
    ud = keep.clone().int()
    ind = torch.randint(0, h * h, (B, S))
    ind[0, S // 2] = ind[0, 0]
    j2d = torch.rand(B, J, 2) * 4 * h
    pix = O.joint_pixels(j2d, h)
    vis = torch.ones(B, J).int()
    total, met, g1, g2, _ = run(m1, m2, None, ind, keep.int(), pix, vis, ud, None, temp, 'channels_last',
                                do_joint=False, gemm_dtype='bf16')
    _, met32, f1, f2, _ = run(m1, m2, None, ind, keep.int(), pix, vis, ud, None, temp, 'channels_last', do_joint=False)
    ld, ad, d1, d2 = O.dense_soft_nce(m1, m2, ind[keep], keep, temp, ud)
    ls, s1, s2, _ = O.scl(m1, m2, j2d, temp, ud, None)
    assert torch.allclose(met[0:2], ld, rtol=1e-2, atol=1e-4), (met[0:2], ld)
    assert float((met[2:4] - ad).abs().max()) <= 2.0 / (int(keep.sum()) * S) + 1e-6
    assert abs(float(met[8]) - float(ls)) < 1e-2 * abs(float(ls)) + 1e-4
    assert rel_l2(g1, d1 + s1) < 2e-2 and rel_l2(g2, d2 + s2) < 2e-2, (rel_l2(g1, d1 + s1), rel_l2(g2, d2 + s2))
    # it really is a different arithmetic (not the fp32 path under another name) ...
    assert not torch.equal(g1, f1)
    # ... and deterministic
    again = run(m1, m2, None, ind, keep.int(), pix, vis, ud, None, temp, 'channels_last', do_joint=False,
                gemm_dtype='bf16')
    assert torch.equal(again[1], met) and torch.equal(again[2], g1) and torch.equal(again[3], g2)


@pytest.mark.parametrize('planes,hi,ho', [(36, 32, 64), (7, 8, 64), (18 * 3, 16, 32), (5, 4, 8), (3, 5, 12)])
@pytest.mark.parametrize('relu', [False, True])
def test_upsample_add_and_its_masked_backward_equal_the_separate_ops(planes, hi, ho, relu):
    """hcm_upsample_bilinear2d_add == (acc + hcm_upsample_bilinear2d) [+ relu] bit for bit, and
    hcm_upsample_bilinear2d_backward_relu == threshold_backward followed by hcm_upsample_bilinear2d_backward."""
    import ctypes as C
    from hcmoco_amd import _lib
    from hcmoco_amd.hip_ops import check
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(planes * 100 + hi)
    x = torch.randn(planes, hi, hi, generator=g).to(dev)
    acc = torch.randn(planes, ho, ho, generator=g).to(dev)
    go = torch.randn(planes, ho, ho, generator=g).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    up = torch.empty_like(acc)
    check(L.hcm_upsample_bilinear2d(p(x), planes, hi, hi, ho, ho, p(up), st), 'up')
    ref = acc + up
    if relu:
        ref = torch.relu(ref)
    out = torch.empty_like(acc)
    check(L.hcm_upsample_bilinear2d_add(p(x), p(acc), int(relu), planes, hi, hi, ho, ho, p(out), st), 'up_add')
    assert torch.equal(out, ref)
    if relu:
        gm_ref = torch.where(ref > 0, go, torch.zeros_like(go))
        dx_ref = torch.empty_like(x)
        check(L.hcm_upsample_bilinear2d_backward(p(gm_ref), planes, hi, hi, ho, ho, p(dx_ref), st), 'bwd')
        dx, gm = torch.empty_like(x), torch.empty_like(go)
        check(L.hcm_upsample_bilinear2d_backward_relu(p(go), p(out), planes, hi, hi, ho, ho, p(dx), p(gm), st), 'bwd_relu')
        assert torch.equal(gm, gm_ref) and torch.equal(dx, dx_ref)
