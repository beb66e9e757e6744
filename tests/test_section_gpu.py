"""The fused loss section (csrc/section.hip, hip_ops.stage2_section) on the MI355X:
every new kernel against a plain-torch float64 statement of the reference op it replaces, the device pixel sampler
bit-exact against the oracle's Philox restatement, and the whole node against the module-by-module path it
replaces.  Tolerances (fp32 kernels vs float64 torch): 1e-5 relative on values, 1e-4 relative L2 on gradients."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import hcmoco_oracle as O  # noqa: E402


def dev():
    return torch.device('cuda:0')


def ops():
    from hcmoco_amd import hip_ops
    return hip_ops


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    n = float(b.norm())
    return float((a - b).norm()) / n if n > 0 else float(a.abs().max())


def make_maps(B, width, size, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, width * 2 ** i, size // 2 ** i, size // 2 ** i, generator=g) for i in range(4)]


@pytest.mark.parametrize('B,width,size,J,F_', [(4, 18, 16, 17, 128), (3, 32, 8, 13, 128), (2, 18, 20, 16, 64),
                                               (32, 18, 64, 17, 128)])
def test_heads_forward_and_backward_against_torch(B, width, size, J, F_):
    """build_backbone.py:265-288: pool + cat + Linear + Normalize for three heads; backward incl. dW/db/dX."""
    torch.manual_seed(B * 100 + width)
    d = dev()
    m1, m2 = make_maps(B, width, size, 1), make_maps(B, width, size, 2)
    Ctot = 15 * width
    feat3 = torch.randn(B, J, 128)
    W = [torch.randn(F_, Ctot) * 0.05, torch.randn(F_, Ctot) * 0.05, torch.randn(F_, 128) * 0.05]
    b = [torch.randn(F_) * 0.1 for _ in range(3)]
    index = torch.randint(0, 2 ** 40, (B,))
    g = lambda t: t.to(d)
    pooled, mean3, ypre, f, fT = ops().heads_forward([g(t) for t in m1], [g(t) for t in m2], g(feat3), g(W[0]), g(b[0]),
                                                     g(W[1]), g(b[1]), g(W[2]), g(b[2]), g(index))
    dd = lambda t: t.double().requires_grad_(True)
    m1d, m2d, f3d = [dd(t) for t in m1], [dd(t) for t in m2], dd(feat3)
    Wd, bd = [dd(t) for t in W], [dd(t) for t in b]
    fref = O.heads(m1d, m2d, f3d, Wd, bd)
    assert f.shape == (B, 3 * F_ + 2)
    assert torch.allclose(f[:, :3 * F_].cpu().double(), fref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(f[:, 3 * F_:].contiguous().view(torch.int64).view(-1).cpu(), index)     # packed all-gather row
    for h in range(3):
        assert torch.equal(fT[h], f[:, h * F_:(h + 1) * F_])
    assert torch.allclose(pooled[0].cpu().double(), torch.cat([t.mean((2, 3)) for t in m1d], 1).detach(), rtol=1e-5, atol=1e-6)
    # backward, with a device-side scale
    gf = torch.randn(3, B, F_)
    gj = torch.randn(B, J, 128)
    scale = torch.tensor(0.37)
    out = ops().heads_backward(g(gf), g(scale), pooled, mean3, ypre, g(W[0]), g(W[1]), g(W[2]), g(gj), J)
    dW1, db1, dW2, db2, dW3, db3, dpooled, gfeat3 = out
    loss = 0.37 * (fref * torch.cat([gf[0], gf[1], gf[2]], 1).double()).sum() + 0.37 * (f3d * gj.double()).sum()
    loss.backward()
    for got, ref in ((dW1, Wd[0].grad), (dW2, Wd[1].grad), (dW3, Wd[2].grad), (db1, bd[0].grad), (db2, bd[1].grad),
                     (db3, bd[2].grad), (gfeat3, f3d.grad)):
        assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)
    # the pooled gradient spread over the maps = the maps' gradient
    hw = [t.shape[2] * t.shape[3] for t in m1]
    off = 0
    for i, t in enumerate(m1d):
        C = t.shape[1]
        ref = t.grad[:, :, 0, 0] * hw[i]
        assert rel_l2(dpooled[0][:, off:off + C], ref) < 1e-4
        off += C
    # bit-repeatable
    out2 = ops().heads_backward(g(gf), g(scale), pooled, mean3, ypre, g(W[0]), g(W[1]), g(W[2]), g(gj), J)
    assert all(torch.equal(a, b_) for a, b_ in zip(out, out2))


@pytest.mark.parametrize('B,H,h,S,J', [(4, 256, 64, 400, 17), (3, 320, 80, 400, 16), (2, 64, 16, 7, 13), (5, 100, 25, 33, 17)])
def test_pixel_sampler_is_bit_exact_against_the_oracle(B, H, h, S, J):
    """contrast_trainer.py:671-685 + :757-761 in one launch; Philox inverse-CDF on integer prefix counts."""
    torch.manual_seed(H + S)
    d = dev()
    mask = (torch.rand(B, H, H) < 0.3).float()
    mask[1] = 0                                    # an image without depth: dropped
    if B > 3:
        mask[3] = 0
        mask[3, H - 1, H - 1] = 1                  # a single valid pixel (nearest resize may or may not keep it)
        mask[3, 0, 0] = 1
    ud = (mask.sum((1, 2)) > 0).long()
    j2d = torch.rand(B, J, 2) * H * 1.3 - 0.15 * H
    for seed, offset, use_depth in ((5, 1, ud), (2 ** 63 + 11, (1 << 63) | 7, None), (9, 3, torch.zeros(B, dtype=torch.long))):
        pix, coord, keep = ops().pixel_sample(mask.to(d), h, h, S, None if use_depth is None else use_depth.to(d),
                                              j2d.to(d), seed, offset)
        ind, keep_ref = O.pixel_sample_philox(mask, h, h, S, use_depth, seed, offset)
        assert torch.equal(keep.cpu().bool(), keep_ref)
        assert torch.equal(coord.cpu(), ind)
        assert torch.equal(pix[:, :S].cpu(), ind)
        assert torch.equal(pix[:, S:].cpu(), O.joint_pixels(j2d, h))
        m = O.nearest_resize_mask(mask, h, h).reshape(B, h * h)
        if bool(keep_ref.any()):
            assert bool((m.gather(1, ind)[keep_ref] > 0).all())        # only valid pixels are ever drawn
    # the oracle's resize rule is the reference's F.interpolate(mode='nearest')
    assert torch.equal(O.nearest_resize_mask(mask, h, h), F.interpolate(mask[:, None], size=(h, h), mode='nearest')[:, 0])
    # statistics: uniform over the valid pixels
    big, _, _ = ops().pixel_sample(mask[:1].to(d), h, h, 200000 if h * h * 4 <= 150 * 1024 else S, None, j2d[:1].to(d), 1, 2)
    if big.shape[1] > 100000:
        valid = O.nearest_resize_mask(mask[:1], h, h).reshape(-1) > 0
        cnt = torch.bincount(big[0, :200000].cpu(), minlength=h * h).double()
        assert float(cnt[~valid].sum()) == 0
        p = cnt[valid] / 200000
        assert abs(float(p.mean()) - 1.0 / int(valid.sum())) < 1e-9 and float(p.std()) < 4.0 * (1.0 / int(valid.sum())) ** 0.5 / 200000 ** 0.5 + 1e-4


def _project_full(maps, W, b):
    size = maps[0].shape[-2:]
    up = [maps[0]] + [F.interpolate(m, size=size, mode='bilinear', align_corners=False) for m in maps[1:]]
    return F.conv2d(torch.cat(up, 1), W, b)


@pytest.mark.parametrize('B,width,size,R', [(3, 18, 16, 37), (2, 32, 8, 20), (32, 18, 64, 417),
                                            # finest maps of 224 / 288 / 320 / 384 crops: the coarsest branch has 49 / 81 /
                                            # 100 / 144 pixels (no divisor of the 256-thread workgroup), the 320 crop's third
                                            # one 400 (a partial second tile); datasets/dataset.py:475 trains at 320
                                            (2, 18, 56, 60), (2, 18, 72, 417), (3, 18, 80, 416), (2, 32, 80, 417),
                                            (2, 18, 96, 100), (1, 18, 40, 33), (1, 18, 24, 9)])
def test_sampled_merge_projection_and_its_backward_against_torch(B, width, size, R):
    """merge_all_res + 1x1 conv (build_backbone.py:243-254) restricted to sampled pixels == gathering the full
    projected map; backward (branch_grad) == torch autograd of that flow plus the average-pool gradient."""
    torch.manual_seed(width + R)
    d = dev()
    m1, m2 = make_maps(B, width, size, 3), make_maps(B, width, size, 4)
    Ctot, Fd = 15 * width, 128
    Wp = [torch.randn(Fd, Ctot, 1, 1) * 0.05 for _ in range(2)]
    bp = [torch.randn(Fd) * 0.1 for _ in range(2)]
    pix = torch.randint(0, size * size, (B, R))
    pix[:, 1] = pix[:, 0]                           # duplicates
    pix[0, 2] = 0
    pix[0, 3] = size * size - 1                     # corners (clamped stencils)
    g = lambda t: t.to(d)
    xs, Wpad, grows = ops().sample_branches([g(t) for t in m1], [g(t) for t in m2], g(pix), g(Wp[0]), g(bp[0]),
                                            g(Wp[1]), g(bp[1]))
    assert float(grows.abs().max()) == 0
    rows = torch.bmm(xs, Wpad.transpose(1, 2)).view(2, B, R, Fd)
    dd = lambda t: t.double().requires_grad_(True)
    md = [[dd(t) for t in m1], [dd(t) for t in m2]]
    Wd, bd = [dd(t) for t in Wp], [dd(t) for t in bp]
    full = [_project_full(md[k], Wd[k], bd[k]) for k in range(2)]
    for k in range(2):
        ref = full[k].flatten(2).gather(2, pix[:, None, :].expand(B, Fd, R)).transpose(1, 2)      # [B, R, F]
        assert torch.allclose(rows[k].cpu().double(), ref.detach(), rtol=1e-5, atol=1e-5)
    # backward
    gr = torch.randn(2, B * R, Fd)
    dpooled = torch.randn(2, B, Ctot)
    scale = torch.tensor(1.7)
    dxs = torch.bmm(g(gr), Wpad)
    dWpad = torch.bmm(g(gr).transpose(1, 2), xs)
    shapes = [tuple(t.shape) for t in m1]
    g1, g2, dWp1, dbp1, dWp2, dbp2 = ops().branch_grad(dxs, g(dpooled), g(scale), g(pix), shapes, dWpad, Fd)
    loss = 0
    for k in range(2):
        sel = full[k].flatten(2).gather(2, pix[:, None, :].expand(B, Fd, R)).transpose(1, 2).reshape(B * R, Fd)
        loss = loss + 1.7 * (sel * gr[k].double()).sum()
        off = 0
        for t in md[k]:
            C = t.shape[1]
            loss = loss + (t.mean((2, 3)) * dpooled[k][:, off:off + C].double()).sum()
            off += C
    loss.backward()
    for k, gm in enumerate((g1, g2)):
        for i in range(4):
            assert rel_l2(gm[i], md[k][i].grad) < 1e-4, (k, i, rel_l2(gm[i], md[k][i].grad))
    for got, ref in ((dWp1, Wd[0].grad), (dWp2, Wd[1].grad), (dbp1, bd[0].grad), (dbp2, bd[1].grad)):
        assert rel_l2(got.reshape(ref.shape), ref) < 1e-4
    # deterministic, and every element is written (poisoned output buffers would show)
    again = ops().branch_grad(dxs, g(dpooled), g(scale), g(pix), shapes, dWpad, Fd)
    for a, b_ in zip(g1 + g2, again[0] + again[1]):
        assert torch.equal(a, b_) and bool(torch.isfinite(a).all())


@pytest.mark.parametrize('B,width,size,R', [(3, 18, 16, 37), (2, 32, 8, 20), (32, 18, 64, 417), (2, 18, 56, 60),
                                            (2, 18, 72, 417), (3, 18, 80, 416), (2, 32, 80, 417), (2, 48, 32, 50),
                                            (2, 18, 96, 100), (1, 18, 40, 33), (1, 18, 24, 9), (56, 18, 80, 416)])
def test_row8_on_the_matrix_cores_against_torch(B, width, size, R):
    """csrc/rowproj.hip (r04): hcm_project_rows == gathering the full-resolution merge_all_res + 1x1 conv map
    (build_backbone.py:243-254, :290-300); hcm_project_rows_backward == torch autograd of that flow plus the
    average-pool gradient, for the branch maps, the projection weights and biases.  fp32 MFMA vs float64: 1e-5 on the
    rows, 1e-4 relative L2 on every gradient.  Sizes: powers of two, the maps of 224 / 288 / 320 / 384 crops, all three
    HRNet widths, the recipe's B = 56 at 320.  Bit-repeatable; every output element written."""
    torch.manual_seed(width + R)
    d = dev()
    m1, m2 = make_maps(B, width, size, 3), make_maps(B, width, size, 4)
    Ctot, Fd = 15 * width, 128
    Wp = [torch.randn(Fd, Ctot, 1, 1) * 0.05 for _ in range(2)]
    bp = [torch.randn(Fd) * 0.1 for _ in range(2)]
    pix = torch.randint(0, size * size, (B, R))
    pix[:, 1] = pix[:, 0]                           # duplicates
    pix[0, 2] = 0
    pix[0, 3] = size * size - 1                     # corners (clamped stencils)
    S = R // 2
    keep = torch.ones(B, dtype=torch.int32)
    if B > 1:
        keep[1] = 0
        pix[1, :S] = 0                              # a dropped image: its dense samples sit on pixel 0, gradient zero
    g = lambda t: t.to(d)
    gm1, gm2 = [g(t) for t in m1], [g(t) for t in m2]
    rows, xs, grows = ops().project_rows(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]))
    assert float(grows.abs().max()) == 0 and rows.shape == (2, B * R, Fd)
    rows_only, none_xs, _ = ops().project_rows(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]), save=False)
    assert none_xs is None and torch.equal(rows_only, rows)
    # r06: hcm_project_rows_cl reads channels-last copies of the branches it gathers from global memory; it must give the
    # same rows and the same saved tile as the NCHW plane walk BIT FOR BIT (same values into the same sums).  (Measured and
    # not the default: hip_ops.ROW8_CHANNELS_LAST.)
    rows_p, xs_p, _ = ops().project_rows(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]),
                                         channels_last=not ops().ROW8_CHANNELS_LAST)
    assert torch.equal(rows_p, rows) and torch.equal(xs_p, xs)
    big = B * size * size * Ctot > 40e6              # float64 autograd of the full maps on the CPU: keep it bounded
    dd = (lambda t: t.double().requires_grad_(True)) if not big else (lambda t: t.float().requires_grad_(True))
    md = [[dd(t) for t in m1], [dd(t) for t in m2]]
    Wd, bd = [dd(t) for t in Wp], [dd(t) for t in bp]
    full = [_project_full(md[k], Wd[k], bd[k]) for k in range(2)]
    sel = []
    for k in range(2):
        ref = full[k].flatten(2).gather(2, pix[:, None, :].expand(B, Fd, R)).transpose(1, 2)      # [B, R, F]
        sel.append(ref)
        assert torch.allclose(rows[k].view(B, R, Fd).cpu().to(ref.dtype), ref.detach(), rtol=1e-5 if not big else 1e-4,
                              atol=1e-5 if not big else 1e-4)
    # the saved rows: [x | 1 | 0]
    assert torch.equal(xs[:, :, Ctot], torch.ones_like(xs[:, :, Ctot])) and float(xs[:, :, Ctot + 1:].abs().max()) == 0
    # backward
    gr = torch.randn(2, B, R, Fd)
    gr[:, keep == 0, :S] = 0                        # what the loss kernels leave for a dropped image
    dpooled = torch.randn(2, B, Ctot)
    scale = torch.tensor(1.7)
    shapes = [tuple(t.shape) for t in m1]
    args = (g(gr).view(2, B * R, Fd), xs, g(Wp[0]), g(Wp[1]), g(dpooled), g(scale), g(pix), shapes, g(keep), S)
    g1, g2, dWp1, dbp1, dWp2, dbp2 = ops().project_rows_backward(*args)
    loss = 0
    for k in range(2):
        loss = loss + 1.7 * (sel[k] * gr[k].to(sel[k].dtype)).sum()
        off = 0
        for t in md[k]:
            C = t.shape[1]
            loss = loss + (t.mean((2, 3)) * dpooled[k][:, off:off + C].to(t.dtype)).sum()
            off += C
    loss.backward()
    tol = 1e-4 if not big else 1e-3
    for k, gm in enumerate((g1, g2)):
        for i in range(4):
            assert rel_l2(gm[i], md[k][i].grad) < tol, (k, i, rel_l2(gm[i], md[k][i].grad))
    for got, ref in ((dWp1, Wd[0].grad), (dWp2, Wd[1].grad), (dbp1, bd[0].grad), (dbp2, bd[1].grad)):
        assert rel_l2(got.reshape(ref.shape), ref) < tol, rel_l2(got.reshape(ref.shape), ref)
    # the same gradients without the keep hint (the dropped image's rows are listed and contribute exact zeros)
    h1, h2 = ops().project_rows_backward(*args[:8], None, 0)[:2]
    for a, b_ in zip(g1 + g2, h1 + h2):
        assert rel_l2(a, b_) < 1e-6
    # deterministic, and every element is written (poisoned output buffers would show)
    again = ops().project_rows_backward(*args)
    for a, b_ in zip(g1 + g2 + [dWp1, dbp1, dWp2, dbp2], again[0] + again[1] + list(again[2:])):
        assert torch.equal(a, b_) and bool(torch.isfinite(a).all())
    # and the r03 path (staging matrix + library GEMMs + owner-computes scan) agrees
    xs0, Wpad, _ = ops().sample_branches(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]))
    assert torch.allclose(xs0, xs, rtol=1e-6, atol=1e-6)      # same taps; the two kernels contract the four products differently
    old = ops().branch_grad(torch.bmm(g(gr).view(2, B * R, Fd), Wpad), g(dpooled), g(scale), g(pix), shapes,
                            torch.bmm(g(gr).view(2, B * R, Fd).transpose(1, 2), xs0), Fd, g(keep), S)
    for a, b_ in zip(g1 + g2 + [dWp1, dbp1, dWp2, dbp2], old[0] + old[1] + list(old[2:])):
        assert rel_l2(a, b_) < 1e-5, rel_l2(a, b_)


@pytest.mark.parametrize('R,hub', [(417, 400), (700, 660)])
def test_row8_backward_with_a_pixel_sampled_hundreds_of_times(R, hub):
    """A mask with a single valid pixel puts all S samples of every image on it (contrast_trainer.py:685 draws with
    replacement): the per-pixel entry lists are then 400 long.  r03's list construction was quadratic there (736 us at
    the bench size); the sorted plan is linear -- and still exact.  660 entries on one pixel are more than a workgroup of
    the finest branch's rows-first path (r06: finest_tiles) stages in LDS: that workgroup takes the generic tile path."""
    torch.manual_seed(5)
    d = dev()
    B, width, size = 4, 18, 64
    m1, m2 = make_maps(B, width, size, 3), make_maps(B, width, size, 4)
    Ctot, Fd = 15 * width, 128
    Wp = [torch.randn(Fd, Ctot, 1, 1) * 0.05 for _ in range(2)]
    bp = [torch.randn(Fd) * 0.1 for _ in range(2)]
    pix = torch.full((B, R), 1234)
    pix[:, hub:] = torch.randint(0, size * size, (B, R - hub))
    g = lambda t: t.to(d)
    rows, xs, grows = ops().project_rows([g(t) for t in m1], [g(t) for t in m2], g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]))
    gr = torch.randn(2, B * R, Fd)
    shapes = [tuple(t.shape) for t in m1]
    g1, g2 = ops().project_rows_backward(g(gr), xs, g(Wp[0]), g(Wp[1]), None, None, g(pix), shapes)[:2]
    md = [[t.double().requires_grad_(True) for t in m1], [t.double().requires_grad_(True) for t in m2]]
    loss = 0
    for k in range(2):
        full = _project_full(md[k], Wp[k].double(), bp[k].double())
        sel = full.flatten(2).gather(2, pix[:, None, :].expand(B, Fd, R)).transpose(1, 2).reshape(B * R, Fd)
        loss = loss + (sel * gr[k].double()).sum()
    loss.backward()
    for k, gm in enumerate((g1, g2)):
        for i in range(4):
            assert rel_l2(gm[i], md[k][i].grad) < 1e-4


def test_range_checked_draw_and_strided_update():
    """memory/mem_bank.py:CMCMem3: the kernels clamp out-of-range rows and raise a sticky device flag; the update reads
    column slices of ONE gathered matrix (row stride 386) without copies."""
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    d = dev()
    torch.manual_seed(0)
    n, K, B = 512, 31, 6
    mem = CMCMem3(128, n, K, 0.07, 0.5, seed=3).to(d)
    y = torch.tensor([5, 7, n + 3, 9, -2, 11], device=d)
    idx = mem.draw(y)
    assert idx[:, 0].tolist() == [5, 7, n - 1, 9, 0, 11]
    ref = O.alias_draw_philox(mem.multinomial.prob.cpu(), mem.multinomial.alias.cpu(), B * (K + 1), mem.multinomial.seed, 0)
    assert torch.equal(idx[:, 1:].cpu(), ref.view(B, K + 1)[:, 1:])
    with pytest.raises(IndexError):
        mem.check_indices()
    mem.check_indices()                             # the flag was consumed
    packed = torch.randn(B, 386, device=d)
    before = [b.clone() for b in mem.banks()]
    yy = torch.tensor([5, 7, 8, 9, 5, 11], device=d)           # duplicate 5: the last one wins
    mem.update_strided([packed[:, 0:128], packed[:, 128:256], packed[:, 256:384]], 386, yy)
    mem.check_indices()
    for i, b in enumerate(mem.banks()):
        ref = O.bank_update(before[i].cpu(), packed[:, 128 * i:128 * (i + 1)].cpu().contiguous(), yy.cpu(), 0.5)
        assert torch.allclose(b.cpu(), ref, rtol=1e-6, atol=1e-7)
    mem.update_strided([packed[:, 0:128], packed[:, 128:256], packed[:, 256:384]], 386, torch.tensor([1, 2, 3, 4, 5, n], device=d))
    with pytest.raises(IndexError):
        mem.check_indices()


def _build(width=18, B=4, size=64, K=256, n=1024, J=17):
    import argparse
    from hcmoco_amd.pycontrast.networks.build_backbone import build_model
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    from hcmoco_amd.pycontrast.datasets.synthetic import SyntheticContrastData
    torch.manual_seed(0)
    opt = argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                             in_channel_list=[3, 3], linear_feat_map=1, width=width, pool_method='mean',
                             skeleton_meta_name={17: 'coco17', 16: 'mpii', 13: 'coco_reduce'}[J], IN_Pretrain=None,
                             depth_Pretrain=None, mem='bank')
    model, _ = build_model(opt)
    model.to(dev()).train()
    mem = CMCMem3(128, n, K, 0.07, 0.5, seed=11).to(dev())
    data = SyntheticContrastData(n, B, size=size, joints=J, steps=1, device=dev(), pool=1)
    return model, mem, data.pool[0]


@pytest.mark.parametrize('stage2,size', [(True, 64), (False, 64), (True, 160), (False, 160), (True, 224)])
def test_fused_section_equals_the_module_path(stage2, size):
    """One autograd node (heads + bank + sampling + projection + three losses) against the module-by-module path it
    replaces (the model's own pooling / Linear / Normalize modules in torch, engine.bank, engine.fmap_sampled) on the
    SAME branch maps, negatives and pixels (the encoders run once: MIOpen may pick another algorithm on a second
    run, which moves the maps by 1e-6): f and losses to 1e-5, every gradient to 1e-4 relative L2; and the node is
    bit-identical run to run."""
    from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
    model, mem, batch = _build(size=size)           # 160 / 224 crops: coarsest maps of 25 / 49 pixels (5^2, 7^2)
    x, index, skel, j2d, vis, ud, mask = batch[0], batch[1], batch[2], batch[4], batch[5], batch[6], batch[7]
    eng = HipLossEngine()
    S, temp = 50, 0.07
    banks0 = [b.clone() for b in mem.banks()]
    idx = mem.draw(index)
    h = x.shape[-1] // 4
    sample_ind, keep = eng.dense_samples(mask, h, h, S, ud)
    model.defer_projection = model.defer_heads = True
    with torch.no_grad():
        f1, f2, f3, f, _ = model(x, skel, return_fm=True)
    assert f is None
    model.defer_projection = model.defer_heads = False
    tail = ([model.head1[0], model.head2[0], model.head3[0]] +
            ([model.encoder1_linear, model.encoder2_linear] if stage2 else []))
    params = [p for l in tail for p in (l.weight, l.bias)]

    def leaves():
        for b, b0 in zip(mem.banks(), banks0):
            b.copy_(b0)
        for p in params:
            p.grad = None
        return ([t.detach().clone().requires_grad_(True) for t in f1], [t.detach().clone().requires_grad_(True) for t in f2],
                f3.detach().clone().requires_grad_(True))

    def finish(total, losses, accs, meters, m1, m2, m3, fvals):
        total.backward()
        torch.cuda.synchronize()
        return (total.detach().clone(), losses.clone(), accs.clone(), meters.clone(), [t.grad.clone() for t in m1 + m2 + [m3]],
                [None if p.grad is None else p.grad.clone() for p in params], [b.clone() for b in mem.banks()],
                fvals.detach().clone())

    def run_fused():
        m1, m2, m3 = leaves()
        tape = {}
        total, losses, accs, meters = eng.section(model, m1, m2, m3, index, mem, stage2, depth_mask=mask, joints2d=j2d,
                                                  joints_vis=vis, use_depth=ud, use_rgb=None, num_samples=S,
                                                  temperature=temp, idx=idx, sample_ind=sample_ind, keep=keep, tape=tape)
        return finish(total, losses, accs, meters, m1, m2, m3, tape['f'])

    def run_modules():
        m1, m2, m3 = leaves()
        fv = torch.cat((model.head1(model._pool(m1)), model.head2(model._pool(m2)), model.head3(m3.mean(1))), dim=1)
        fa, fb, fc = torch.chunk(fv, 3, dim=1)
        total, losses, accs = eng.bank(mem, fa, fb, fc, index, fa, fb, fc, index, use_depth=ud, use_rgb=None, idx=idx)
        meters = torch.zeros(9, device=dev())
        if stage2:
            fm_total, meters = eng.fmap_sampled(m1, m2, model.encoder1_linear, model.encoder2_linear, m3, mask, j2d, vis,
                                                ud, None, S, temp, sample_ind=sample_ind, keep=keep)
            total = total + fm_total
        return finish(total, losses, accs, meters, m1, m2, m3, fv)

    a = run_fused()
    b = run_modules()
    assert torch.allclose(a[7], b[7], rtol=1e-5, atol=1e-6), float((a[7] - b[7]).abs().max())      # f
    assert abs(float(a[0]) - float(b[0])) <= 1e-5 * abs(float(b[0]))
    assert torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6) and torch.allclose(a[2], b[2], atol=1e-3)
    if stage2:
        assert torch.allclose(a[3], b[3], rtol=1e-5, atol=1e-6), (a[3], b[3])
    for ga, gb in zip(a[4], b[4]):
        assert rel_l2(ga, gb) < 1e-4, rel_l2(ga, gb)
    for pa, pb in zip(a[5], b[5]):
        assert pa is not None and pb is not None and rel_l2(pa, pb) < 1e-4, rel_l2(pa, pb)
    for ba, bb in zip(a[6], b[6]):
        assert torch.allclose(ba, bb, rtol=1e-5, atol=1e-6)
    # the node is bit-repeatable: totals, meters, every gradient it produces
    a2 = run_fused()
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and torch.equal(a[3], a2[3])
    for ga, gb in zip(a[4] + a[5], a2[4] + a2[5]):
        assert torch.equal(ga, gb)


def test_section_draws_its_own_negatives_and_pixels_and_counts_launches():
    """Product mode (nothing injected): negatives by hcm_alias_draw_checked, pixels by hcm_pixel_sample; the tape shows
    valid draws; the Philox offsets advance."""
    from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
    model, mem, batch = _build()
    x, index, skel, j2d, vis, ud, mask = batch[0], batch[1], batch[2], batch[4], batch[5], batch[6], batch[7]
    eng = HipLossEngine()
    model.defer_projection = model.defer_heads = True
    outs = []
    for _ in range(2):
        f1, f2, f3, f, aux = model(x, skel, return_fm=True)
        tape = {}
        total, losses, accs, meters = eng.section(model, f1, f2, f3, index, mem, True, depth_mask=mask, joints2d=j2d,
                                                  joints_vis=vis, use_depth=ud, use_rgb=None, num_samples=40,
                                                  temperature=0.07, tape=tape)
        total.backward()
        outs.append(tape)
    torch.cuda.synchronize()
    h = x.shape[-1] // 4
    m = O.nearest_resize_mask(mask.cpu(), h, h).reshape(mask.shape[0], -1)
    for t in outs:
        assert torch.equal(t['idx'][:, 0], index)
        keepb = t['keep'].bool().cpu()
        assert torch.equal(keepb, (m.sum(1) > 0) & bool(ud.sum() > 0))
        assert bool((m.gather(1, t['coord'].cpu())[keepb] > 0).all())
    assert not torch.equal(outs[0]['idx'][:, 1:], outs[1]['idx'][:, 1:])
    assert not torch.equal(outs[0]['coord'], outs[1]['coord'])
    assert bool(torch.isfinite(total))
    model.defer_projection = model.defer_heads = False


def test_channels_last_is_refused_with_the_encoder_runtime():
    """args.channels_last was an r01 experiment for stock ATen encoders; the encoder runtime and the
    section kernels are NCHW, so the trainer says so instead of failing inside the first convolution."""
    import tempfile
    import bench
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    args = bench.make_args(4, 512, 1024, 64, 'coco17', 'nccl', tempfile.mkdtemp(), 2)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, True
    tr = ContrastTrainer(args)
    tr.device = dev()
    with pytest.raises(ValueError, match='channels_last needs the stock ATen encoders'):
        bench.build(args, tr, dev())


def test_ranks_bind_to_the_cores_of_their_gpus_numa_node():
    """learning/affinity.py on the real box: the GPU's PCI address resolves to a NUMA node and a cpu list in sysfs; binding
    leaves the process on a non-empty subset of that list (restored afterwards), HCM_PIN_NUMA=0 leaves it alone."""
    import os
    from hcmoco_amd.pycontrast.learning import affinity
    info = affinity.gpu_node(0)
    if info is None:
        pytest.skip('no NUMA information for this device in sysfs')
    node, cpus = info
    assert node >= 0 and len(cpus) > 0
    before = os.sched_getaffinity(0)
    try:
        assert affinity.pin_to_gpu_node(0, mode='0') is None and os.sched_getaffinity(0) == before
        got = affinity.pin_to_gpu_node(0, mode='node')
        assert got is not None and got[0] == node
        now = os.sched_getaffinity(0)
        assert now and now <= set(cpus) and now <= before
        share = affinity.pin_to_gpu_node(0, mode='share')          # one visible GPU: the whole node again
        assert share is not None and set(share[1]) <= set(cpus)
    finally:
        os.sched_setaffinity(0, before)
