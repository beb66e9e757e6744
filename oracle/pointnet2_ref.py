"""ctypes binding of oracle/_ref/libpointnet2_ref_{ieee,fma}.so = the REFERENCE's PointNet++ kernels compiled for
gfx950 by oracle/build_ref_pointnet2.py (TEST INFRASTRUCTURE ONLY; needs a GPU).  `CONTRACT` selects the build:
'fma' (contraction on, scalar: what nvcc's default gives the reference) or 'ieee' (`-ffp-contract=off`).

Mirrors the reference's pybind module `pointnet2_cuda` (networks/pointnet2/src/pointnet2_api.cpp:10-24): the same
nine `*_wrapper` names with the same argument order as its `*_wrapper_fast` functions (ball_query.cpp:16-28,
group_points.cpp, sampling.cpp:12-52, interpolate.cpp), each forwarding to the kernel launcher on torch's current
stream -- which is all those wrappers do.  Tensors are the caller's (ROCm, contiguous, float32 / int32)."""
import ctypes as C
import json
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SOS = {k: os.path.join(_HERE, '_ref', 'libpointnet2_ref_%s.so' % k) for k in ('ieee', 'fma')}
SYMS = os.path.join(_HERE, '_ref', 'pointnet2_ref_symbols.json')
CONTRACT = 'fma'
_libs = {}
_i, _f, _p = C.c_int, C.c_float, C.c_void_p

_SIGS = {
    'ball_query_kernel_launcher_fast': [_i, _i, _i, _f, _i, _p, _p, _p, _p],
    'group_points_kernel_launcher_fast': [_i] * 5 + [_p] * 4,
    'group_points_grad_kernel_launcher_fast': [_i] * 5 + [_p] * 4,
    'gather_points_kernel_launcher_fast': [_i] * 4 + [_p] * 4,
    'gather_points_grad_kernel_launcher_fast': [_i] * 4 + [_p] * 4,
    'furthest_point_sampling_kernel_launcher': [_i] * 3 + [_p] * 4,
    'three_nn_kernel_launcher_fast': [_i] * 3 + [_p] * 5,
    'three_interpolate_kernel_launcher_fast': [_i] * 4 + [_p] * 5,
    'three_interpolate_grad_kernel_launcher_fast': [_i] * 4 + [_p] * 5,
}


def available():
    return all(os.path.exists(p) for p in SOS.values()) and os.path.exists(SYMS)


def _fn(name):
    fns = _libs.get(CONTRACT)
    if fns is None:
        lib = C.CDLL(SOS[CONTRACT])
        with open(SYMS) as f:
            table = json.load(f)
        fns = _libs[CONTRACT] = {}
        for n, sig in _SIGS.items():
            fn = getattr(lib, table[n])
            fn.restype, fn.argtypes = None, sig
            fns[n] = fn
    return fns[name]


def _ptr(t, dtype):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype, (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _fp(t):
    return _ptr(t, torch.float32)


def _ip(t):
    return _ptr(t, torch.int32)


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _fn('ball_query_kernel_launcher_fast')(b, n, m, float(radius), nsample, _fp(new_xyz), _fp(xyz), _ip(idx), _st())
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _fn('group_points_kernel_launcher_fast')(b, c, n, npoints, nsample, _fp(points), _ip(idx), _fp(out), _st())
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _fn('group_points_grad_kernel_launcher_fast')(b, c, n, npoints, nsample, _fp(grad_out), _ip(idx),
                                                   _fp(grad_points), _st())
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _fn('gather_points_kernel_launcher_fast')(b, c, n, npoints, _fp(points), _ip(idx), _fp(out), _st())
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _fn('gather_points_grad_kernel_launcher_fast')(b, c, n, npoints, _fp(grad_out), _ip(idx), _fp(grad_points), _st())
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _fn('furthest_point_sampling_kernel_launcher')(b, n, m, _fp(points), _fp(temp), _ip(idx), _st())
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _fn('three_nn_kernel_launcher_fast')(b, n, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx), _st())


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _fn('three_interpolate_kernel_launcher_fast')(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out), _st())


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _fn('three_interpolate_grad_kernel_launcher_fast')(b, c, n, m, _fp(grad_out), _ip(idx), _fp(weight),
                                                        _fp(grad_points), _st())
