"""Rows 12-17 against the REFERENCE'S OWN KERNELS, compiled for gfx950 (oracle/build_ref_pointnet2.py ->
oracle/_ref/libpointnet2_ref_{ieee,fma}.so, bound by oracle/pointnet2_ref.py): the reference's CUDA sources,
translated by torch's bundled hipify and built with hipcc -O2, run on the MI355X on the same inputs as

  * the HIP ops of this repo (`hcmoco_amd.pointnet2_hip`), and
  * the plain-C restatement (`oracle/pointnet2_oracle.c`),

through the same nine `*_wrapper` calls of the reference's `pointnet2_cuda` module, in BOTH arithmetic contracts:
the reference built with `-ffp-contract=off` against `ieee`, and built with contraction on (scalar, as a target
without packed fp32 math contracts it) against `fma`.  Indices, distances and the gather / group / interpolate
forwards are compared BIT FOR BIT; the three backward kernels of the reference use float atomics (accumulation
order undefined), so they are compared bit for bit on duplicate-free indices and to 1e-5 / 1e-4 otherwise.
This pins the C restatement (and the HIP ops) to the reference for these rows: selection rules, tie-breaks,
initial values, loop bounds and both roundings of the distance expressions.

Skipped (not failed) when the libraries are absent: they are built where /root/reference exists and travel as
prebuilt .so files.
"""
import pytest
import torch

from oracle import pointnet2_oracle as P
from oracle import pointnet2_ref as R

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not R.available(), reason='oracle/_ref/libpointnet2_ref.so not built '
                                                          '(python oracle/build_ref_pointnet2.py where /root/reference exists)')]


def mod():
    import hcmoco_amd.pointnet2_hip as m
    return m


def d():
    return torch.device('cuda:0')


def cloud(B, N, seed, dup=True):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g)
    if dup and N > 8:
        src = torch.randint(0, N, (B, N // 4), generator=g)
        dst = torch.randint(0, N, (B, N // 4), generator=g)
        for b in range(B):
            xyz[b, dst[b]] = xyz[b, src[b]]
    return xyz


def depth_cloud(B, seed):
    """back-projected 256x256 depth map + the 4096 points sampled from it with replacement
    (networks/build_backbone.py:420-455)"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing='ij')
    z = 3.0 + 0.1 * torch.randn(B, 256, 256, generator=g)
    full = torch.stack([(xs - 128) * z * 0.0035, (128 - ys) * z * 0.0035, z - 3.0], -1).reshape(B, 65536, 3).contiguous()
    pick = torch.randint(0, 65536, (B, 4096), generator=g)
    return full, torch.gather(full, 1, pick.unsqueeze(-1).expand(B, 4096, 3)).contiguous()


@pytest.fixture(autouse=True, params=['fma', 'ieee'])
def contract(request):
    """the reference build, the HIP ops and the C restatement are switched together"""
    m = mod()
    old = m.CONTRACT
    m.CONTRACT = R.CONTRACT = request.param
    P.set_contract(request.param)
    yield request.param
    m.CONTRACT = old
    R.CONTRACT = 'fma'
    P.set_contract('fma')


@pytest.mark.parametrize('N,M', [(5, 5), (37, 20), (64, 64), (100, 33), (1024, 256), (1500, 700), (4096, 1024),
                                 (9000, 50), (4096, 4096)])
def test_fps_reference_kernel_vs_hip_vs_c_restatement(N, M):
    B = 3
    xyz = cloud(B, N, N * 7 + M)
    x = xyz.to(d())
    outs = []
    for m in (R, mod()):
        idx = torch.zeros(B, M, dtype=torch.int32, device=d())
        temp = torch.full((B, N), 1e10, device=d())
        m.furthest_point_sampling_wrapper(B, N, M, x, temp, idx)
        outs.append((idx.cpu(), temp.cpu()))
    oi, ot = P.furthest_point_sampling(xyz, M)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # reference == HIP
    assert torch.equal(outs[0][0], oi) and torch.equal(outs[0][1], ot)                     # reference == C restatement


def test_fps_tie_break_of_the_reference_kernel():
    """equidistant points: the reference's reduction tree keeps the candidate of the bit-reversed-lowest thread
    (SURVEY 8a-12 correction) -- the known answer the C restatement was written to, now observed on the kernel."""
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]]], device=d())
    for m in (R, mod()):
        out = torch.zeros(1, 5, dtype=torch.int32, device=d())
        temp = torch.full((1, 5), 1e10, device=d())
        m.furthest_point_sampling_wrapper(1, 5, 5, xyz, temp, out)
        assert out.cpu().tolist() == [[0, 4, 3, 2, 1]]
    # a regular grid: every step is a many-way tie
    g = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(4.), indexing='ij'), -1).reshape(1, 256, 3)
    res = []
    for m in (R, mod()):
        out = torch.zeros(1, 256, dtype=torch.int32, device=d())
        temp = torch.full((1, 256), 1e10, device=d())
        m.furthest_point_sampling_wrapper(1, 256, 256, g.to(d()).contiguous(), temp, out)
        res.append(out.cpu())
    assert torch.equal(res[0], res[1])
    assert torch.equal(res[0], P.furthest_point_sampling(g.contiguous(), 256)[0])


def test_fps_constant_and_two_valued_clouds():
    """The clouds of empty-mask images are all zeros (networks/build_backbone.py:427-445): every round of FPS is a 4096-way
    tie, which in the r04 kernel is the path where every lane of every wave holds the maximum (one ds_max_u64 post per wave
    after a second wave reduction).  Also a cloud of two coincident clusters, and one with a single point that differs."""
    n, m = 4096, 192
    clouds = torch.zeros(3, n, 3)
    clouds[1, n // 3:] = torch.tensor([0.25, -1.0, 2.0])
    clouds[2, 1234] = torch.tensor([1.0, 1.0, 1.0])
    res = []
    for mod_ in (R, mod()):
        out = torch.zeros(3, m, dtype=torch.int32, device=d())
        temp = torch.full((3, n), 1e10, device=d())
        mod_.furthest_point_sampling_wrapper(3, n, m, clouds.to(d()), temp, out)
        res.append((out.cpu(), temp.cpu()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[1][0], P.furthest_point_sampling(clouds, m)[0])


@pytest.mark.parametrize('N,M,r,ns', [(10, 4, 0.3, 3), (300, 300, 0.2, 16), (4096, 1024, 0.125, 32),
                                      (2500, 777, 0.05, 16), (1024, 256, 1.0, 32), (4096, 1024, 0.1, 16)])
def test_ball_query_reference_kernel_vs_hip_vs_c_restatement(N, M, r, ns):
    B = 2
    xyz = cloud(B, N, N + M)
    new_xyz = xyz[:, torch.randperm(N, generator=torch.Generator().manual_seed(N))[:M]].contiguous()
    new_xyz[:, 0] = 50.0
    outs = []
    for m in (R, mod()):
        idx = torch.zeros(B, M, ns, dtype=torch.int32, device=d())
        m.ball_query_wrapper(B, N, M, r, ns, new_xyz.to(d()), xyz.to(d()), idx)
        outs.append(idx.cpu())
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], P.ball_query(r, ns, xyz, new_xyz))


@pytest.mark.parametrize('n,m', [(7, 2), (7, 3), (256, 64), (4096, 1024), (3000, 4096)])
def test_three_nn_reference_kernel_vs_hip_vs_c_restatement(n, m):
    B = 2
    unknown, known = cloud(B, n, n), cloud(B, m, m + 1)
    outs = []
    for mm in (R, mod()):
        dist2 = torch.zeros(B, n, 3, device=d())
        idx = torch.zeros(B, n, 3, dtype=torch.int32, device=d())
        mm.three_nn_wrapper(B, n, m, unknown.to(d()), known.to(d()), dist2, idx)
        outs.append((dist2.cpu(), idx.cpu()))
    rd, ri = P.three_nn(unknown, known)
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], ri) and torch.equal(outs[0][0], rd)


def test_pts2depth_scale_reference_kernel_vs_hip():
    """BASELINE config 4's own size, the FULL problem on both sides (B=8 clouds of 65 536 unknown x 4096 known
    points, then the interpolation the reference chains to it, pointnet2_utils.py:100-150)."""
    B, n, m, Cc = 8, 65536, 4096, 16
    full, known = depth_cloud(B, 2026)
    feats = torch.randn(B, Cc, m, generator=torch.Generator().manual_seed(5)).to(d())
    res = []
    for mm in (R, mod()):
        dist2 = torch.zeros(B, n, 3, device=d())
        idx = torch.zeros(B, n, 3, dtype=torch.int32, device=d())
        mm.three_nn_wrapper(B, n, m, full.to(d()), known.to(d()), dist2, idx)
        w = 1.0 / (torch.sqrt(dist2) + 1e-8)
        w = (w / w.sum(2, keepdim=True)).contiguous()
        out = torch.empty(B, Cc, n, device=d())
        mm.three_interpolate_wrapper(B, Cc, m, n, feats, idx, w, out)
        res.append((dist2.cpu(), idx.cpu(), out.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert int((res[0][0][..., 0] == 0).sum()) > 0
    # and the sampling + grouping of the first set-abstraction level on the same clouds (4096 -> 1024, r=0.1, 16)
    res = []
    for mm in (R, mod()):
        idx = torch.zeros(B, 1024, dtype=torch.int32, device=d())
        temp = torch.full((B, m), 1e10, device=d())
        mm.furthest_point_sampling_wrapper(B, m, 1024, known.to(d()), temp, idx)
        centres = torch.empty(B, 3, 1024, device=d())
        mm.gather_points_wrapper(B, 3, m, 1024, known.to(d()).transpose(1, 2).contiguous(), idx, centres)
        ball = torch.zeros(B, 1024, 16, dtype=torch.int32, device=d())
        mm.ball_query_wrapper(B, m, 1024, 0.1, 16, centres.transpose(1, 2).contiguous(), known.to(d()), ball)
        res.append((idx.cpu(), centres.cpu(), ball.cpu()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_group_gather_interpolate_forward_bit_exact_and_grads():
    g = torch.Generator().manual_seed(1)
    B, C, N, npts, ns = 2, 37, 500, 128, 16
    pts = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, npts, ns), generator=g, dtype=torch.int32)
    go = torch.randn(B, C, npts, ns, generator=g)
    gi = torch.randint(0, N, (B, 77), generator=g, dtype=torch.int32)
    go2 = torch.randn(B, C, 77, generator=g)
    n, m = 900, 200
    feats = torch.randn(B, C, m, generator=g)
    ii = torch.randint(0, m, (B, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    go3 = torch.randn(B, C, n, generator=g)
    # duplicate-free variants: every target row receives at most one contribution -> order cannot matter
    perm = torch.stack([torch.randperm(N, generator=g)[:npts * 3] for _ in range(B)]).to(torch.int32)
    idx_u = perm.view(B, npts, 3).contiguous()
    go_u = torch.randn(B, C, npts, 3, generator=g)
    gi_u = perm[:, :77].contiguous()
    ii_u = torch.stack([torch.randperm(3 * 60, generator=g) for _ in range(B)]).to(torch.int32).view(B, 60, 3).contiguous()
    w_u = torch.rand(B, 60, 3, generator=g)
    go3_u = torch.randn(B, C, 60, generator=g)
    res = []
    for mm in (R, mod()):
        o = {}
        o['group'] = torch.empty(B, C, npts, ns, device=d())
        mm.group_points_wrapper(B, C, N, npts, ns, pts.to(d()), idx.to(d()), o['group'])
        o['group_grad'] = torch.zeros(B, C, N, device=d())
        mm.group_points_grad_wrapper(B, C, N, npts, ns, go.to(d()), idx.to(d()), o['group_grad'])
        o['group_grad_u'] = torch.zeros(B, C, N, device=d())
        mm.group_points_grad_wrapper(B, C, N, npts, 3, go_u.to(d()), idx_u.to(d()), o['group_grad_u'])
        o['gather'] = torch.empty(B, C, 77, device=d())
        mm.gather_points_wrapper(B, C, N, 77, pts.to(d()), gi.to(d()), o['gather'])
        o['gather_grad'] = torch.zeros(B, C, N, device=d())
        mm.gather_points_grad_wrapper(B, C, N, 77, go2.to(d()), gi.to(d()), o['gather_grad'])
        o['gather_grad_u'] = torch.zeros(B, C, N, device=d())
        mm.gather_points_grad_wrapper(B, C, N, 77, go2.to(d()), gi_u.to(d()), o['gather_grad_u'])
        o['interp'] = torch.empty(B, C, n, device=d())
        mm.three_interpolate_wrapper(B, C, m, n, feats.to(d()), ii.to(d()), w.to(d()), o['interp'])
        o['interp_grad'] = torch.zeros(B, C, m, device=d())
        mm.three_interpolate_grad_wrapper(B, C, n, m, go3.to(d()), ii.to(d()), w.to(d()), o['interp_grad'])
        o['interp_grad_u'] = torch.zeros(B, C, 180, device=d())
        mm.three_interpolate_grad_wrapper(B, C, 60, 180, go3_u.to(d()), ii_u.to(d()), w_u.to(d()), o['interp_grad_u'])
        res.append({k: v.cpu() for k, v in o.items()})
    ref, hip = res
    for k in ('group', 'gather', 'interp', 'group_grad_u', 'gather_grad_u', 'interp_grad_u'):
        assert torch.equal(ref[k], hip[k]), k
    assert torch.allclose(ref['group_grad'], hip['group_grad'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ref['gather_grad'], hip['gather_grad'], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ref['interp_grad'], hip['interp_grad'], rtol=1e-4, atol=1e-5)
    # the C restatement against the reference kernels
    assert torch.equal(ref['group'], P.group_points(pts, idx))
    assert torch.equal(ref['gather'], P.gather_points(pts, gi))
    assert torch.equal(ref['interp'], P.three_interpolate(feats, ii, w))
    assert torch.equal(ref['group_grad_u'], P.group_points_grad(go_u, idx_u, N))
    assert torch.equal(ref['gather_grad_u'], P.gather_points_grad(go2, gi_u, N))
    assert torch.equal(ref['interp_grad_u'], P.three_interpolate_grad(go3_u, ii_u, w_u, 180))
    assert torch.allclose(ref['group_grad'], P.group_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)
    assert torch.allclose(ref['interp_grad'], P.three_interpolate_grad(go3, ii, w, m), rtol=1e-4, atol=1e-5)


def test_the_two_reference_builds_differ_where_the_two_contracts_differ(contract):
    """On a cloud with many near-ties the two builds of the reference give different last bits, each equal to
    the restatement in its own mode and to neither in the other."""
    n, m = 2048, 512
    unknown, known = cloud(1, n, 11), cloud(1, m, 12)
    dist2 = torch.zeros(1, n, 3, device=d())
    idx = torch.zeros(1, n, 3, dtype=torch.int32, device=d())
    R.three_nn_wrapper(1, n, m, unknown.to(d()), known.to(d()), dist2, idx)
    other = 'ieee' if contract == 'fma' else 'fma'
    same = P.three_nn(unknown, known)[0]
    P.set_contract(other)
    try:
        diff = P.three_nn(unknown, known)[0]
    finally:
        P.set_contract(contract)
    assert torch.equal(dist2.cpu(), same)
    assert not torch.equal(dist2.cpu(), diff)
