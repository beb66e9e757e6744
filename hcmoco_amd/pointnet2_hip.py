"""Drop-in for the reference's pybind module ``pointnet2_cuda``
(/root/reference/pycontrast/networks/pointnet2/src/pointnet2_api.cpp:10-24).

Same nine function names, same positional arguments, same ownership rules: the CALLER allocates
and pre-initialises every output tensor (see pointnet2_utils.py in the reference); the wrappers
only read device pointers of contiguous fp32/int32 ROCm tensors and launch on the current
stream.  Unlike the reference, a bad argument or a failed launch raises instead of ``exit(-1)``.

    import hcmoco_amd.pointnet2_hip as pointnet2      # instead of: import pointnet2_cuda as pointnet2
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .hip_ops import _dev, _stream


# Arithmetic contract of FPS / ball query / three_nn / three_interpolate (include/hcmoco_hip.h):
# 'fma' = what the reference's `nvcc -O2` build computes (fused multiply-adds, the default),
# 'ieee' = un-fused fp32 (an --fmad=false or CPU build).  HCM_PN2_CONTRACT=ieee selects the latter.
import os as _os

CONTRACTS = {'ieee': 0, 'fma': 1}
CONTRACT = _os.environ.get('HCM_PN2_CONTRACT', 'fma')


def _contract():
    return CONTRACTS[CONTRACT]


def _f(t, name):
    return _dev(t, torch.float32, name)


def _i(t, name):
    return _dev(t, torch.int32, name)


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    check(_lib.lib().hcm_ball_query_contract(b, n, m, float(radius), nsample, _f(new_xyz, 'ball_query'),
                                             _f(xyz, 'ball_query'), _i(idx, 'ball_query'), _contract(), _stream()),
          'hcm_ball_query')
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    check(_lib.lib().hcm_group_points(b, c, n, npoints, nsample, _f(points, 'group_points'),
                                      _i(idx, 'group_points'), _f(out, 'group_points'), _stream()),
          'hcm_group_points')
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    check(_lib.lib().hcm_group_points_grad(b, c, n, npoints, nsample, _f(grad_out, 'group_points_grad'),
                                           _i(idx, 'group_points_grad'), _f(grad_points, 'group_points_grad'),
                                           _stream()), 'hcm_group_points_grad')
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    check(_lib.lib().hcm_gather_points(b, c, n, npoints, _f(points, 'gather_points'), _i(idx, 'gather_points'),
                                       _f(out, 'gather_points'), _stream()), 'hcm_gather_points')
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    check(_lib.lib().hcm_gather_points_grad(b, c, n, npoints, _f(grad_out, 'gather_points_grad'),
                                            _i(idx, 'gather_points_grad'), _f(grad_points, 'gather_points_grad'),
                                            _stream()), 'hcm_gather_points_grad')
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    check(_lib.lib().hcm_furthest_point_sampling_contract(b, n, m, _f(points, 'fps'), _f(temp, 'fps'), _i(idx, 'fps'),
                                                          _contract(), _stream()), 'hcm_furthest_point_sampling')
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    check(_lib.lib().hcm_three_nn_contract(b, n, m, _f(unknown, 'three_nn'), _f(known, 'three_nn'), _f(dist2, 'three_nn'),
                                           _i(idx, 'three_nn'), _contract(), _stream()), 'hcm_three_nn')


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    check(_lib.lib().hcm_three_interpolate_contract(b, c, m, n, _f(points, 'three_interpolate'),
                                                    _i(idx, 'three_interpolate'), _f(weight, 'three_interpolate'),
                                                    _f(out, 'three_interpolate'), _contract(), _stream()),
          'hcm_three_interpolate')


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    check(_lib.lib().hcm_three_interpolate_grad(b, c, n, m, _f(grad_out, 'three_interpolate_grad'),
                                                _i(idx, 'three_interpolate_grad'), _f(weight, 'three_interpolate_grad'),
                                                _f(grad_points, 'three_interpolate_grad'), _stream()),
          'hcm_three_interpolate_grad')


# --------------------------------------------------------------------------------------------- #
# LDS-resident backward shared by group_points / gather_points / three_interpolate (an addition of
# this build; `pointnet2_cuda` has no counterpart).  See include/hcmoco_hip.h.
# --------------------------------------------------------------------------------------------- #
LDS_SCATTER_MAX_TARGETS = 140 * 1024 // 4


def deterministic():
    """HCM_DETERMINISTIC=1: no path with floating-point atomics may be taken silently."""
    import os
    return os.environ.get('HCM_DETERMINISTIC', '0') != '0'


def scatter_add_lds(grad_out, idx, coef, m, div=1):
    """grad_points[b,c,j] = sum_{q: idx[b,q]==j} coef[b,q] * grad_out[b,c,q//div]  -> [B, C, m]."""
    B, Cc, qsrc = grad_out.shape
    idx2 = idx.reshape(B, -1).contiguous()
    out = torch.empty(B, Cc, m, dtype=torch.float32, device=grad_out.device)
    cf = C.c_void_p(0) if coef is None else _f(coef.reshape(B, -1).contiguous(), 'scatter_add_lds')
    check(_lib.lib().hcm_scatter_add_lds(_f(grad_out, 'scatter_add_lds'), cf, _i(idx2, 'scatter_add_lds'), B, Cc, qsrc,
                                         idx2.shape[1], m, div, _f(out, 'scatter_add_lds'), _stream()),
          'hcm_scatter_add_lds')
    return out


def scatter_plan(idx, coef, m, div=1):
    """The streaming order, collision ranks and heavy-target flags of an index tensor idx [B, ...] (values in [0, m)) and
    its weights: what ``scatter_add_planned`` consumes; built once per (idx, coef) and shared by every backward."""
    B = idx.shape[0]
    idx2 = idx.reshape(B, -1).contiguous()
    Q = idx2.shape[1]
    if Q % div:
        raise ValueError('scatter_plan: %d contributions are not a multiple of div=%d' % (Q, div))
    qsrc = Q // div
    n = int(_lib.lib().hcm_scatter_plan_elems(B, qsrc, div, m))
    if n == 0:
        raise ValueError('scatter_plan: unsupported shape B=%d Qsrc=%d div=%d m=%d' % (B, qsrc, div, m))
    plan = torch.empty(n, dtype=torch.int32, device=idx.device)
    pcoef = None if coef is None else torch.empty(n, dtype=torch.float32, device=idx.device)
    cf = C.c_void_p(0) if coef is None else _f(coef.reshape(B, -1).contiguous(), 'scatter_plan')
    check(_lib.lib().hcm_scatter_plan(_i(idx2, 'scatter_plan'), cf, B, qsrc, div, m, _i(plan, 'scatter_plan'),
                                      C.c_void_p(0) if pcoef is None else _f(pcoef, 'scatter_plan'), _stream()),
          'hcm_scatter_plan')
    return plan, pcoef, qsrc


def scatter_add_planned(grad_out, idx, coef, m, div=1):
    """grad_points[b,c,j] = sum_{q: idx[b,q]==j} coef[b,q] * grad_out[b,c,q//div] -> [B, C, m], deterministic (source
    order, per-wave channel ownership, ranked LDS read-add-write rounds, no atomics).  The plan is cached on the index
    tensor, keyed by the weights it was built with."""
    B, Cc, qsrc = grad_out.shape
    key = (idx._version, m, div, None if coef is None else (coef.data_ptr(), coef._version))
    cached = getattr(idx, '_hcm_plan', None)
    if cached is None or cached[0] != key:
        cached = (key, scatter_plan(idx, coef, m, div))
        idx._hcm_plan = cached
    plan, pcoef, q = cached[1]
    if q != qsrc:
        raise ValueError('scatter_add_planned: grad_out has %d sources, the index tensor %d' % (qsrc, q))
    out = torch.empty(B, Cc, m, dtype=torch.float32, device=grad_out.device)
    check(_lib.lib().hcm_scatter_add_planned(_f(grad_out, 'scatter_add_planned'), _i(plan, 'scatter_add_planned'),
                                             C.c_void_p(0) if pcoef is None else _f(pcoef, 'scatter_add_planned'), B, Cc,
                                             qsrc, m, div, _f(out, 'scatter_add_planned'), _stream()),
          'hcm_scatter_add_planned')
    return out


class _BallProject(torch.autograd.Function):
    """y = relu?(batchnorm(P[b, :, idx[b, i, j]] + Wxyz D[b, :, i, j])) -> [B, C, np, ns]: the first SharedMLP layer on the
    implicit grouped tensor (hcm_ball_project_*, csrc/bnact.hip).  Backward returns dP (planned scatter of dz,
    deterministic), dWxyz, dgamma, dbeta; the offsets D carry no gradient."""

    @staticmethod
    def forward(ctx, P, D, Wxyz, idx, gamma, beta, running_mean, running_var, momentum, eps, relu):
        for t in (D, Wxyz, gamma, beta) + (() if P is None else (P,)):
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError('hcmoco_amd.ball_project needs fp32 ROCm tensors (no CPU fallback exists)')
        D, Wxyz, idx = D.contiguous(), Wxyz.contiguous(), idx.contiguous()
        B, three, npnt, ns = D.shape
        Cc = Wxyz.shape[0]
        N = 0
        if P is not None:
            P = P.contiguous()
            N = P.shape[2]
        L = _lib.lib()
        nf = int(L.hcm_ball_project_stats_floats(B, Cc, npnt, ns))
        if (nf == 0 or three != 3 or tuple(Wxyz.shape) != (Cc, 3) or tuple(idx.shape) != (B, npnt, ns)
                or (P is not None and tuple(P.shape[:2]) != (B, Cc))):
            raise ValueError('ball_project: unsupported shape P %s D %s Wxyz %s idx %s' % (
                None if P is None else tuple(P.shape), tuple(D.shape), tuple(Wxyz.shape), tuple(idx.shape)))
        y = torch.empty(B, Cc, npnt, ns, dtype=torch.float32, device=D.device)
        stats = torch.empty(nf, dtype=torch.float32, device=D.device)
        rm = C.c_void_p(0) if running_mean is None else _f(running_mean, 'ball_project')
        rv = C.c_void_p(0) if running_var is None else _f(running_var, 'ball_project')
        pp = C.c_void_p(0) if P is None else _f(P, 'ball_project')
        check(L.hcm_ball_project_forward(pp, _f(D, 'ball_project'), _f(Wxyz, 'ball_project'), _i(idx, 'ball_project'),
                                         _f(gamma, 'ball_project'), _f(beta, 'ball_project'), rm, rv, float(momentum),
                                         float(eps), int(bool(relu)), B, Cc, N, npnt, ns, _f(y, 'ball_project'),
                                         _f(stats, 'ball_project'), _stream()), 'hcm_ball_project_forward')
        ctx.save_for_backward(D, Wxyz, gamma, stats, y, *(() if P is None else (P,)))
        ctx.idx, ctx.relu, ctx.has_p = idx, bool(relu), P is not None   # the same idx OBJECT: the scatter plan is cached on it
        return y

    @staticmethod
    def backward(ctx, dy):
        D, Wxyz, gamma, stats, y = ctx.saved_tensors[:5]
        P = ctx.saved_tensors[5] if ctx.has_p else None
        idx = ctx.idx
        B, _, npnt, ns = D.shape
        Cc = Wxyz.shape[0]
        N = P.shape[2] if P is not None else 0
        dy = dy.contiguous()
        # without point features (the first SA level) nothing consumes dz: the entry point then makes ONE pass over (dy, y, D)
        # and writes the parameter gradients only (the 134 / 537 MB dz tensors of that level are never allocated)
        dz = torch.empty_like(y) if P is not None else None
        dW = torch.empty_like(Wxyz)
        gstats = torch.empty_like(stats)
        pp = C.c_void_p(0) if P is None else _f(P, 'ball_project')
        check(_lib.lib().hcm_ball_project_backward(_f(dy, 'ball_project'), _f(y, 'ball_project'), pp, _f(D, 'ball_project'),
                                                   _f(Wxyz, 'ball_project'), _i(idx, 'ball_project'), _f(gamma, 'ball_project'),
                                                   _f(stats, 'ball_project'), int(ctx.relu), B, Cc, N, npnt, ns,
                                                   C.c_void_p(0) if dz is None else _f(dz, 'ball_project'),
                                                   _f(dW, 'ball_project'), _f(gstats, 'ball_project'),
                                                   _stream()), 'hcm_ball_project_backward')
        dP = None
        if P is not None and ctx.needs_input_grad[0]:
            if N <= LDS_SCATTER_MAX_TARGETS:
                dP = scatter_add_planned(dz.view(B, Cc, npnt * ns), idx, None, N, 1)
            elif deterministic():
                raise RuntimeError('ball_project backward: %d source points exceed the planned scatter (%d) and '
                                   'HCM_DETERMINISTIC forbids the atomic fallback' % (N, LDS_SCATTER_MAX_TARGETS))
            else:                                   # a target axis too long for LDS: ATen's scatter (atomics)
                dP = torch.zeros_like(P).scatter_add_(2, idx.view(B, 1, -1).long().expand(B, Cc, -1), dz.view(B, Cc, -1))
        return dP, None, dW, None, gstats[:Cc], gstats[Cc:2 * Cc], None, None, None, None, None


def ball_project(P, D, Wxyz, idx, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=True):
    """The first conv -> BatchNorm2d -> ReLU of a PointNet++ SharedMLP applied to the grouped tensor WITHOUT building it:
    P [B, C1, N] = W_f features (or None), D [B, 3, np, ns] = xyz[idx] - centre (``ball_offsets``), Wxyz [C1, 3], idx
    [B, np, ns] the ball members."""
    if D.requires_grad:
        raise RuntimeError('ball_project: the coordinates carry no gradient on this path (detach D)')
    return _BallProject.apply(P, D, Wxyz, idx, gamma, beta, running_mean, running_var, momentum, eps, relu)


def ball_project_supported(B, Cc, npnt, ns):
    """Shapes the hcm_ball_project_* kernels take (ball of 4..64 members, npoint a multiple of 4)."""
    return int(_lib.lib().hcm_ball_project_stats_floats(B, Cc, npnt, ns)) != 0


class _PointProject(torch.autograd.Function):
    """P[b] = W src[b] for W [K, C], src [B, C, N] -> [B, K, N] on csrc/conv1x1.hip (hcm_conv1x1_forward / _backward_data /
    hcm_conv1x1_ball_wgrad): the source-point projection of ``Conv2d.forward_grouped``.  Its weight gradient is a [K x C] output
    with a reduction over all B * N points -- as a library GEMM a handful of workgroups walking 0.5 M columns each (0.4 ms
    per call in profiles/r05_hrnetpn_timeline.txt); the ball kernel splits the points over the chip and sums the partial
    blocks in fixed order."""

    @staticmethod
    def forward(ctx, W, src):
        W, src = W.contiguous(), src.contiguous()
        B, Cs, N = src.shape
        K = W.shape[0]
        out = torch.empty(B, K, N, dtype=torch.float32, device=src.device)
        check(_lib.lib().hcm_conv1x1_forward_exact(_f(src, 'point_project'), _f(W, 'point_project'), _f(out, 'point_project'), B, Cs, K,
                                             N, _stream()), 'hcm_conv1x1_forward_exact')
        ctx.save_for_backward(W, src)
        return out

    @staticmethod
    def backward(ctx, dP):
        W, src = ctx.saved_tensors
        B, Cs, N = src.shape
        K = W.shape[0]
        dP = dP.contiguous()
        L = _lib.lib()
        dW = dsrc = None
        if ctx.needs_input_grad[1]:
            dsrc = torch.empty_like(src)
            check(L.hcm_conv1x1_backward_data_exact(_f(dP, 'point_project'), _f(W, 'point_project'), _f(dsrc, 'point_project'), B, Cs, K,
                                              N, _stream()), 'hcm_conv1x1_backward_data_exact')
        if ctx.needs_input_grad[0]:
            need = int(L.hcm_conv1x1_ball_wgrad_workspace_bytes(B, Cs, K, N, 1))
            ws = torch.empty(need // 4, dtype=torch.float32, device=src.device)
            dW = torch.empty_like(W)
            check(L.hcm_conv1x1_ball_wgrad_exact(_f(src, 'point_project'), _f(dP, 'point_project'), B, Cs, K, N, 1, _f(dW, 'point_project'),
                                           _f(ws, 'point_project'), need, _stream()), 'hcm_conv1x1_ball_wgrad_exact')
        return dW, dsrc


def set_conv1x1_arith(exact):
    """Arithmetic of hcm_conv1x1_forward / _backward_data / _ball_wgrad (csrc/conv1x1.hip), process-wide: ``exact=False`` (default)
    = layers of 64+ channels on the bf16 matrix cores with split operands (three terms, fp32 accumulate, 4.4e-6 of float64),
    ``exact=True`` = exact fp32 MFMA everywhere.  Returns the previous setting."""
    prev = _lib.lib().hcm_conv1x1_set_arith(1 if exact else 0)
    if prev < 0:
        raise RuntimeError('hcm_conv1x1_set_arith failed')
    return bool(prev)


def point_project_supported(K, Cs, N):
    """Shapes ``point_project`` takes: K and C multiples of 16 (the caller pads the source channels), N a multiple of 256."""
    return K % 16 == 0 and Cs % 16 == 0 and N % 256 == 0 and N > 0


def point_project(W, src):
    """W [K, C] applied to every point of src [B, C, N] (fp32, ROCm) -> [B, K, N]; see ``_PointProject``."""
    if not (W.is_cuda and src.is_cuda and W.dtype == src.dtype == torch.float32):
        raise RuntimeError('hcmoco_amd.point_project needs fp32 ROCm tensors (no CPU fallback exists)')
    if W.dim() != 2 or src.dim() != 3 or W.shape[1] != src.shape[1] or not point_project_supported(W.shape[0], src.shape[1], src.shape[2]):
        raise ValueError('point_project: unsupported shapes W %s src %s' % (tuple(W.shape), tuple(src.shape)))
    return _PointProject.apply(W, src)


class _BallMax(torch.autograd.Function):
    """max over the last axis with ATen max_pool2d's tie rule (first index); hcm_rowmax_*."""

    @staticmethod
    def forward(ctx, x):
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hcmoco_amd.ball_max needs fp32 ROCm tensors (no CPU fallback exists)')
        x = x.contiguous()
        ns = x.shape[-1]
        rows = x.numel() // ns
        y = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
        arg = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
        check(_lib.lib().hcm_rowmax_forward(_f(x, 'ball_max'), rows, ns, _f(y, 'ball_max'), _i(arg, 'ball_max'), _stream()),
              'hcm_rowmax_forward')
        ctx.save_for_backward(arg)
        ctx.ns = ns
        return y

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty(*arg.shape, ctx.ns, dtype=torch.float32, device=g.device)
        check(_lib.lib().hcm_rowmax_backward(_f(g, 'ball_max'), _i(arg, 'ball_max'), arg.numel(), ctx.ns,
                                             _f(dx, 'ball_max'), _stream()), 'hcm_rowmax_backward')
        return dx


def ball_max(x):
    """x [..., nsample] -> max over the last axis (the reference's F.max_pool2d(x, [1, nsample]).squeeze(-1))."""
    return _BallMax.apply(x)
