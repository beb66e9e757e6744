"""Known-answer tests pinning oracle/pointnet2_oracle.c (the reference ships no fixtures for its
CUDA kernels).  Hand-checkable tiny clouds.  CPU only."""
import pytest
import torch

from oracle import pointnet2_oracle as P


def test_fps_collinear_and_ties():
    # points on a line at x = 0,1,2,3,10 : start 0 -> farthest 10 (idx 4) -> then 3? dist to {0,10}:
    # (squared) x=1:1, x=2:4, x=3:9 -> idx 3 ; then x=1:min(1,4)=1, x=2:min(4,1)=1 -> a tie
    xyz = torch.tensor([[[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]]])
    idx, temp = P.furthest_point_sampling(xyz, 5)
    # bs = 4 threads, thread t owns k = t, t+4.  The tree (sampling_gpu.cu:140-200) folds slot
    # t+s into slot t keeping slot t on ties: level s=2 moves thread 2's candidate (k=2, d=1) into
    # slot 0 (thread 0 holds d=0), level s=1 keeps slot 0 over slot 1 (k=1, d=1) -> k=2 wins.
    # i.e. ties go to the smaller BIT-REVERSED thread id, not to the smaller thread id.
    assert idx.tolist() == [[0, 4, 3, 2, 1]]
    # temp holds min-distances to the picks made BEFORE the last round (the last pick is never applied)
    assert temp[0].tolist() == [0, 1, 0, 0, 0]


def test_fps_duplicates_tie_break_depends_on_virtual_thread():
    # n=6 -> bs=4. duplicates of the far point at k=1 (thread 1) and k=4 (thread 0): thread 0 wins
    xyz = torch.zeros(1, 6, 3)
    xyz[0, 1, 0] = 5
    xyz[0, 4, 0] = 5
    idx, _ = P.furthest_point_sampling(xyz, 2)
    assert idx.tolist() == [[0, 4]]
    # all points identical: every distance is 0 > -1, first strict max of thread 0 is k=0
    idx, _ = P.furthest_point_sampling(torch.ones(2, 7, 3), 4)
    assert idx.tolist() == [[0, 0, 0, 0], [0, 0, 0, 0]]


def test_ball_query_first_hits_pad_and_empty():
    xyz = torch.tensor([[[0., 0, 0], [0.1, 0, 0], [5, 0, 0], [0.2, 0, 0], [0.05, 0, 0]]])
    centres = torch.tensor([[[0., 0, 0], [5, 0, 0], [100, 0, 0]]])
    idx = P.ball_query(0.15, 3, xyz, centres)
    assert idx[0, 0].tolist() == [0, 1, 4]        # first three in index order (0.2 is outside)
    assert idx[0, 1].tolist() == [2, 2, 2]        # one hit pads every slot
    assert idx[0, 2].tolist() == [0, 0, 0]        # no hit: caller's zeros survive
    # strict '<' : a point exactly on the sphere is outside
    idx = P.ball_query(0.5, 2, torch.tensor([[[0.5, 0, 0], [0.25, 0, 0]]]), torch.zeros(1, 1, 3))
    assert idx[0, 0].tolist() == [1, 1]
    # n < nsample
    idx = P.ball_query(1.0, 4, torch.tensor([[[0.1, 0, 0], [0.2, 0, 0]]]), torch.zeros(1, 1, 3))
    assert idx[0, 0].tolist() == [0, 1, 0, 0]


def test_group_gather_and_grads():
    pts = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    idx = torch.tensor([[[3, 3], [0, 1]], [[2, 2], [2, 0]]], dtype=torch.int32)
    out = P.group_points(pts, idx)
    assert out[0, 1].tolist() == [[7, 7], [4, 5]] and out[1, 2].tolist() == [[22, 22], [22, 20]]
    g = P.group_points_grad(torch.ones(2, 3, 2, 2), idx, 4)
    assert g[0, 0].tolist() == [1, 1, 0, 2] and g[1, 0].tolist() == [1, 0, 3, 0]
    gi = torch.tensor([[1, 1, 0]], dtype=torch.int32)
    assert P.gather_points(pts[:1], gi)[0, 0].tolist() == [1, 1, 0]
    assert P.gather_points_grad(torch.tensor([[[1., 2, 4]]]), gi, 4)[0, 0].tolist() == [4, 3, 0, 0]


def test_three_nn_and_interpolate():
    known = torch.tensor([[[0., 0, 0], [1, 0, 0], [1, 0, 0], [3, 0, 0]]])
    unknown = torch.tensor([[[0.9, 0, 0], [10, 0, 0]]])
    d2, idx = P.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [1, 2, 0]          # duplicate distance: first wins, strict '<'
    assert torch.allclose(d2[0, 0], torch.tensor([0.01, 0.01, 0.81]), atol=1e-6)
    assert idx[0, 1].tolist() == [3, 1, 2]
    # fewer than 3 known points: unset slots keep index 0 and the 1e40 tracker -> +inf as float
    d2, idx = P.three_nn(unknown[:, :1], known[:, :2])
    assert idx[0, 0].tolist() == [1, 0, 0] and d2[0, 0, 2] == float('inf')
    feats = torch.tensor([[[1., 2, 4, 8]]])
    w = torch.tensor([[[0.5, 0.25, 0.25], [1.0, 0, 0]]])
    ii = torch.tensor([[[1, 2, 0], [3, 1, 2]]], dtype=torch.int32)
    out = P.three_interpolate(feats, ii, w)
    assert out[0, 0].tolist() == [0.5 * 2 + 0.25 * 4 + 0.25 * 1, 8.0]
    g = P.three_interpolate_grad(torch.tensor([[[2., 3]]]), ii, w, 4)
    assert g[0, 0].tolist() == [0.5, 1.0, 0.5, 3.0]


def test_arithmetic_contracts_are_distinguishable():
    """The two contracts of oracle/pointnet2_oracle.c on inputs where they differ by construction.
    fma mode: fma(dz, dz, fma(dx, dx, dy*dy)) -- only dy*dy is rounded; ieee mode rounds every product."""
    import numpy as np
    f = np.float32
    dx, dy, dz = f(1.0) + f(2.0 ** -12), f(0.0), f(0.0)        # dx*dx = 1 + 2^-11 + 2^-24: inexact in fp32
    ieee = f(dx * dx)                                          # rounds to 1 + 2^-11
    known = torch.tensor([[[0.0, 0, 0]]])
    unknown = torch.tensor([[[float(dx), 0, 0]]])
    try:
        P.set_contract('ieee')
        assert P.get_contract() == 'ieee'
        d_ieee, _ = P.three_nn(unknown, known)
        P.set_contract('fma')
        d_fma, _ = P.three_nn(unknown, known)
        # with dy = dz = 0 both give round(dx*dx): the contracts agree when only one term is non-zero
        assert float(d_ieee[0, 0, 0]) == float(ieee) == float(d_fma[0, 0, 0])
        # two non-zero terms: fma keeps dx*dx exact inside the fused add, ieee rounds it first
        a = f(1.0) + f(2.0 ** -12)
        b = f(2.0 ** -12)                                      # dy*dy = 2^-24 exactly
        unknown = torch.tensor([[[float(a), float(b), 0]]])
        P.set_contract('ieee')
        d_ieee, _ = P.three_nn(unknown, known)
        P.set_contract('fma')
        d_fma, _ = P.three_nn(unknown, known)
        exact = float(np.float64(a) * np.float64(a) + np.float64(b) * np.float64(b))   # 1 + 2^-11 + 2^-23
        want_ieee = f(f(a * a) + f(b * b))                     # (1 + 2^-11) + 2^-24 -> ties-to-even -> 1 + 2^-11
        want_fma = f(exact)                                    # one rounding of the exact sum -> 1 + 2^-11 + 2^-23
        assert float(d_ieee[0, 0, 0]) == float(want_ieee)
        assert float(d_fma[0, 0, 0]) == float(want_fma)
        assert float(want_fma) != float(want_ieee)
        # three_interpolate: w0*p0 + w1*p1 + w2*p2, same contraction shape
        feats = torch.tensor([[[float(a), float(b), 0.0]]])
        w = torch.tensor([[[float(a), float(b), 0.0]]])
        ii = torch.tensor([[[0, 1, 2]]], dtype=torch.int32)
        P.set_contract('ieee')
        o_ieee = float(P.three_interpolate(feats, ii, w)[0, 0, 0])
        P.set_contract('fma')
        o_fma = float(P.three_interpolate(feats, ii, w)[0, 0, 0])
        assert o_ieee == float(want_ieee) and o_fma == float(want_fma)
    finally:
        P.set_contract('fma')



def test_reference_kernel_build_exports_the_nine_launchers():
    """oracle/_ref (the reference's own kernels for gfx950, oracle/build_ref_pointnet2.py): where the reference is
    present the recipe must have produced both builds; anywhere, a present build must carry every launcher
    oracle/pointnet2_ref.py binds.  (No device code runs here; the GPU comparison is tests/test_pointnet2_ref_gpu.py.)"""
    import json
    import os
    import subprocess
    from oracle import build_ref_pointnet2 as B
    from oracle import pointnet2_ref as R
    if os.path.isdir(B.REF_SRC):
        assert B.build() is not None and R.available()
    if not R.available():
        pytest.skip('oracle/_ref not built and no reference to build it from')
    with open(R.SYMS) as f:
        table = json.load(f)
    assert sorted(table) == sorted(B.LAUNCHERS) == sorted(R._SIGS)
    for so in R.SOS.values():
        nm = subprocess.run(['nm', '-D', '--defined-only', so], capture_output=True, text=True, check=True).stdout
        for sym in table.values():
            assert sym in nm, (so, sym)
