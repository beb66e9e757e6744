"""ctypes loader for oracle/pointnet2_oracle.c (TEST INFRASTRUCTURE ONLY).

Functions take/return CPU torch tensors and follow the reference's Python wrappers
(/root/reference/pycontrast/networks/pointnet2/pointnet2_utils.py) for output allocation and
pre-initialisation (idx zeros, temp 1e10, grads zeros)."""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libpointnet2_oracle.so')
_lib = None


def build():
    res = subprocess.run(['make', '-C', _HERE], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('oracle build failed:\n' + res.stdout + res.stderr)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def set_contract(mode):
    """'fma' (the reference's nvcc -O2 build, default) or 'ieee' (un-fused fp32)."""
    lib().oracle_set_contract({'ieee': 0, 'fma': 1}[mode])


def get_contract():
    return 'fma' if lib().oracle_get_contract() else 'ieee'


def furthest_point_sampling(xyz, npoint):
    B, N, _ = xyz.shape
    xyz = xyz.contiguous().float()
    out = torch.zeros(B, npoint, dtype=torch.int32)
    temp = torch.full((B, N), 1e10, dtype=torch.float32)
    lib().oracle_furthest_point_sampling(B, N, npoint, _p(xyz), _p(temp), _p(out))
    return out, temp


def ball_query(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, nsample, dtype=torch.int32)
    a, b = new_xyz.contiguous().float(), xyz.contiguous().float()
    lib().oracle_ball_query(B, N, M, C.c_float(radius), nsample, _p(a), _p(b), _p(idx))
    return idx


def group_points(points, idx):
    B, Cc, N = points.shape
    _, npts, ns = idx.shape
    out = torch.zeros(B, Cc, npts, ns, dtype=torch.float32)
    points, idx = points.contiguous(), idx.contiguous()
    lib().oracle_group_points(B, Cc, N, npts, ns, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, N):
    B, Cc, npts, ns = grad_out.shape
    g = torch.zeros(B, Cc, N, dtype=torch.float32)
    grad_out, idx = grad_out.contiguous(), idx.contiguous()
    lib().oracle_group_points_grad(B, Cc, N, npts, ns, _p(grad_out), _p(idx), _p(g))
    return g


def gather_points(points, idx):
    B, Cc, N = points.shape
    M = idx.shape[1]
    out = torch.zeros(B, Cc, M, dtype=torch.float32)
    points, idx = points.contiguous(), idx.contiguous()
    lib().oracle_gather_points(B, Cc, N, M, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, N):
    B, Cc, M = grad_out.shape
    g = torch.zeros(B, Cc, N, dtype=torch.float32)
    grad_out, idx = grad_out.contiguous(), idx.contiguous()
    lib().oracle_gather_points_grad(B, Cc, N, M, _p(grad_out), _p(idx), _p(g))
    return g


def three_nn(unknown, known):
    """Returns (dist2 SQUARED, idx); the reference wrapper applies sqrt afterwards (pointnet2_utils.py:98)."""
    B, N, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.zeros(B, N, 3, dtype=torch.float32)
    idx = torch.zeros(B, N, 3, dtype=torch.int32)
    u, k = unknown.contiguous().float(), known.contiguous().float()
    lib().oracle_three_nn(B, N, m, _p(u), _p(k), _p(dist2), _p(idx))
    return dist2, idx


def three_nn_rows(unknown, known, rows):
    """three_nn for the unknown points ``rows`` (1-D index tensor) of every batch element only -- the same
    per-point scan, used where the full problem (32 x 65536 x 4096) would take minutes on the host."""
    return three_nn(unknown[:, rows].contiguous(), known)


def three_interpolate(points, idx, weight):
    B, Cc, m = points.shape
    n = idx.shape[1]
    out = torch.zeros(B, Cc, n, dtype=torch.float32)
    points, idx, weight = points.contiguous(), idx.contiguous(), weight.contiguous()
    lib().oracle_three_interpolate(B, Cc, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    B, Cc, n = grad_out.shape
    g = torch.zeros(B, Cc, m, dtype=torch.float32)
    grad_out, idx, weight = grad_out.contiguous(), idx.contiguous(), weight.contiguous()
    lib().oracle_three_interpolate_grad(B, Cc, n, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g
