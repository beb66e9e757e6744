"""ctypes binding of libhcmoco_hip.so (the C ABI declared in include/hcmoco_hip.h).

Only plain pointers and sizes cross this boundary: callers pass ``tensor.data_ptr()`` and the
raw ``hipStream_t`` of the current torch stream.  There is deliberately NO fallback: a missing
library is an ImportError-class failure, a non-zero return code raises ``HipError``."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(CSRC, 'libhcmoco_hip.so')
GLUE_PATH = os.path.join(CSRC, 'libhcmoco_torch.so')
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'hcmoco_hip.h')


class HipError(RuntimeError):
    pass


class Strides4(C.Structure):
    _fields_ = [('sN', C.c_int64), ('sC', C.c_int64), ('sH', C.c_int64), ('sW', C.c_int64)]


_p, _i, _i64, _u64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t


class Branches(C.Structure):
    """hcm_branches / hcm_branches_out: the four NCHW maps of one HRNet (include/hcmoco_hip.h)."""
    _fields_ = [('map', C.c_void_p * 4), ('C', C.c_int * 4), ('H', C.c_int * 4), ('W', C.c_int * 4)]


# name -> (restype, argtypes); mirrors include/hcmoco_hip.h one to one
SIGNATURES = {
    # *_bf16 twins are added below the table (same argument lists, uint16 bank pointers)
    'hcm_abi_version': (_i, []),
    'hcm_error_string': (C.c_char_p, [_i]),
    'hcm_alias_build': (_i, [_p, _i64, _p, _p]),
    'hcm_alias_draw': (_i, [_p, _p, _i64, _p, _i, _i, _u64, _u64, _p, _p]),
    'hcm_bank_nce_workspace_bytes': (_sz, [_i, _i, _i]),
    'hcm_bank_nce_fused': (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f,
                                _p, _p, _p, _p, _p, _p, _sz, _p]),
    'hcm_bank_nce_fused_timed': (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f,
                                      _p, _p, _p, _p, _p, _p, _sz, _p, _i, _p]),
    'hcm_bank_logits_fwd': (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p]),
    'hcm_bank_logits_bwd': (_i, [_p, _p, _p, _i64, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    'hcm_bank_update': (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _f, _p]),
    'hcm_moco_logits': (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p]),
    'hcm_moco_enqueue': (_i, [_p, _p, _i, _i, _i, _i64, _p]),
    'hcm_dense_soft_nce_workspace_bytes': (_sz, [_i, _i, _i]),
    'hcm_dense_soft_nce': (_i, [_p, _p, Strides4, _i, _i, _i, _i, _p, _p, _i, _f, _p, _p, _p, _p, _sz, _p]),
    'hcm_dense_soft_nce_coords': (_i, [_p, _p, Strides4, _i, _i, _i, _i, _p, _p, _i, _p, _i, _f, _p, _p, _p, _p, _sz, _p]),
    'hcm_sample_rows': (_i, [_p, Strides4, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p]),
    'hcm_sample_rows_grad': (_i, [_p, _i, _i, Strides4, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    'hcm_sampling_matrix': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'hcm_joint_nce_workspace_bytes': (_sz, [_i, _i, _i]),
    'hcm_joint_nce': (_i, [_p, _p, Strides4, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f,
                           _p, _p, _p, _p, _p, _sz, _p]),
    'hcm_scl_workspace_bytes': (_sz, [_i, _i, _i]),
    'hcm_scl': (_i, [_p, _p, Strides4, _i, _i, _i, _i, _p, _p, _p, _i, _f, _p, _p, _p, _p, _sz, _p]),
    'hcm_joint_pixels': (_i, [_p, _i, _i, _p, _p]),
    'hcm_upsample_bilinear2d': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'hcm_upsample_bilinear2d_backward': (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    'hcm_upsample_bilinear2d_nhwc': (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    'hcm_furthest_point_sampling': (_i, [_i, _i, _i, _p, _p, _p, _p]),
    'hcm_ball_query': (_i, [_i, _i, _i, _f, _i, _p, _p, _p, _p]),
    'hcm_group_points': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'hcm_group_points_grad': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'hcm_gather_points': (_i, [_i, _i, _i, _i, _p, _p, _p, _p]),
    'hcm_gather_points_grad': (_i, [_i, _i, _i, _i, _p, _p, _p, _p]),
    'hcm_three_nn': (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    'hcm_three_interpolate': (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'hcm_three_interpolate_grad': (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'hcm_furthest_point_sampling_contract': (_i, [_i, _i, _i, _p, _p, _p, _i, _p]),
    'hcm_ball_query_contract': (_i, [_i, _i, _i, _f, _i, _p, _p, _p, _i, _p]),
    'hcm_three_nn_contract': (_i, [_i, _i, _i, _p, _p, _p, _p, _i, _p]),
    'hcm_three_interpolate_contract': (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    'hcm_scatter_add_lds': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    'hcm_scatter_plan_elems': (_sz, [_i] * 4),
    'hcm_scatter_plan': (_i, [_p, _p] + [_i] * 4 + [_p] * 3),
    'hcm_scatter_add_planned': (_i, [_p] * 3 + [_i] * 5 + [_p, _p]),
    'hcm_sgc_workspace_floats': (C.c_size_t, [_i] * 4),
    'hcm_sgc_forward': (_i, [_p] * 12 + [_i] * 7 + [_f, _f] + [_p] * 6),
    'hcm_sgc_backward': (_i, [_p] * 12 + [_i] * 7 + [_p] * 7),
    'hcm_bn_act_stats_floats': (C.c_size_t, [_i, _i, _i]),
    'hcm_bn_act_forward': (_i, [_p] * 6 + [_f, _f] + [_i] * 4 + [_p] * 3),
    'hcm_bn_act_backward': (_i, [_p] * 6 + [_i] * 4 + [_p] * 4),
    'hcm_upsample_bilinear2d_add': (_i, [_p, _p] + [_i] * 6 + [_p, _p]),
    'hcm_upsample_bilinear2d_backward_relu': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _p]),
    'hcm_conv_wgrad_partial': (_i, [_i, _p, _p] + [_i] * 5 + [_p, C.c_size_t, _p, _p]),
    'hcm_wgrad_reduce_batch': (_i, [_p, _i, _p]),
    'hcm_conv3x3_stats_slots': (_i, [_i] * 2),
    'hcm_conv3x3_forward_stats': (_i, [_p] * 3 + [_i] * 5 + [_p] * 3),
    'hcm_bn_act_forward_pre': (_i, [_p] * 6 + [_f, _f] + [_i] * 4 + [_p] * 3 + [_i, _p]),
    'hcm_conv3x3_supported': (_i, [_i] * 4),
    'hcm_conv3x3_forward': (_i, [_p] * 3 + [_i] * 5 + [_p]),
    'hcm_conv3x3_backward_data': (_i, [_p] * 3 + [_i] * 5 + [_p]),
    'hcm_bn_act_backward_ws': (_i, [_p] * 6 + [_i] * 4 + [_p] * 5),
    'hcm_conv3x3_wgrad_workspace_bytes': (C.c_size_t, [_i] * 5),
    'hcm_conv3x3_wgrad': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _sz, _p]),
    'hcm_conv1x1_wgrad_workspace_bytes': (C.c_size_t, [_i] * 5),
    'hcm_conv1x1_wgrad': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _sz, _p]),
    'hcm_conv3x3s2_wgrad_workspace_bytes': (C.c_size_t, [_i] * 5),
    'hcm_conv3x3s2_wgrad': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _sz, _p]),
    'hcm_bn_relu_ballmax_stats_floats': (_sz, [_i] * 4),
    'hcm_bn_relu_ballmax_forward': (_i, [_p] * 5 + [_f, _f] + [_i] * 4 + [_p] * 5),
    'hcm_bn_relu_ballmax_backward': (_i, [_p] * 7 + [_i] * 4 + [_p] * 3),
    'hcm_conv1x1_set_arith': (_i, [_i]),
    'hcm_conv1x1_supported': (_i, [_i] * 3),
    'hcm_conv1x1_forward': (_i, [_p] * 3 + [_i] * 4 + [_p]),
    'hcm_conv1x1_backward_data': (_i, [_p] * 3 + [_i] * 4 + [_p]),
    'hcm_conv1x1_ball_wgrad_workspace_bytes': (C.c_size_t, [_i] * 5),
    'hcm_conv1x1_ball_wgrad': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _sz, _p]),
    'hcm_conv1x1_forward_exact': (_i, [_p] * 3 + [_i] * 4 + [_p]),
    'hcm_conv1x1_backward_data_exact': (_i, [_p] * 3 + [_i] * 4 + [_p]),
    'hcm_conv1x1_ball_wgrad_exact': (_i, [_p, _p] + [_i] * 5 + [_p, _p, _sz, _p]),
    'hcm_ball_project_stats_floats': (_sz, [_i] * 4),
    'hcm_ball_project_forward': (_i, [_p] * 8 + [_f, _f] + [_i] * 6 + [_p] * 3),
    'hcm_ball_project_backward': (_i, [_p] * 8 + [_i] * 6 + [_p] * 4),
    'hcm_rowmax_forward': (_i, [_p, C.c_longlong, _i, _p, _p, _p]),
    'hcm_rowmax_backward': (_i, [_p, _p, C.c_longlong, _i, _p, _p]),
    'hcm_heads_forward': (_i, [Branches, Branches, _p] + [_i] * 5 + [_p] * 11 + [_i, _p, _p]),
    'hcm_heads_backward': (_i, [_p] * 5 + [_i] * 5 + [_p] * 14),
    'hcm_pixel_sample': (_i, [_p] + [_i] * 6 + [_p, _p, _i, _u64, _u64, _p, _p, _p, _p]),
    'hcm_sample_branches_ld': (_i, [_i]),
    'hcm_sample_branches': (_i, [Branches, Branches, _i, _p] + [_i] * 3 + [_p] * 8),
    'hcm_branch_grad': (_i, [_p] * 4 + [_i] * 3 + [Branches, Branches, _p, _i, _p, _i] + [_p] * 5),
    'hcm_section_total': (_i, [_p, _p, _p, _p]),
    'hcm_project_rows': (_i, [Branches, Branches, _i, _p] + [_i] * 3 + [_p] * 8),
    'hcm_project_rows_nhwc_floats': (_sz, [Branches, Branches, _i, _i]),
    'hcm_project_rows_cl': (_i, [Branches, Branches, _i, _p] + [_i] * 3 + [_p] * 8 + [_sz, _p]),
    'hcm_project_rows_dw_workspace_bytes': (_sz, [_i, _i]),
    'hcm_project_rows_dw': (_i, [_p] * 3 + [_i] * 4 + [_p] * 5 + [_sz, _p]),
    'hcm_project_rows_backward_workspace_bytes': (_sz, [_i, _i, _i, Branches]),
    'hcm_project_rows_backward': (_i, [_p] * 7 + [_i] * 4 + [Branches, Branches, _p, _i] + [_p] * 5 + [_sz, _p]),
    'hcm_alias_draw_checked': (_i, [_p, _p, _i64, _p, _i, _i, _u64, _u64, _p, _p, _p]),
    'hcm_bank_update_checked': (_i, [_p, _p, _p, _i64, _p, _p, _p, _i64, _p, _i, _i, _f, _p, _p]),
    'hcm_prof_enable': (_i, [_i]),
    'hcm_prof_read': (_i, [_p, _p]),
    'hcm_prof_read_tag': (_i, [_i, _p, _p]),
    'hcm_prof_read_work': (_i, [_i, _p]),
}

SIGNATURES['hcm_dense_soft_nce_coords_bf16'] = SIGNATURES['hcm_dense_soft_nce_coords']
SIGNATURES['hcm_scl_bf16'] = SIGNATURES['hcm_scl']
SIGNATURES['hcm_dense_soft_nce_coords_exact'] = SIGNATURES['hcm_dense_soft_nce_coords']
SIGNATURES['hcm_scl_exact'] = SIGNATURES['hcm_scl']

for _name in ('hcm_bank_nce_fused', 'hcm_bank_nce_fused_timed', 'hcm_bank_logits_fwd', 'hcm_bank_logits_bwd',
              'hcm_bank_update', 'hcm_bank_update_checked'):
    SIGNATURES[_name + '_bf16'] = SIGNATURES[_name]

_lib = None
ABI_VERSION = 6       # HCM_ABI_VERSION of include/hcmoco_hip.h this binding was written against


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libhcmoco_hip.so (in-tree, via make + hipcc)."""
    import torch
    res = subprocess.run(['make', '-C', CSRC, '-j', '4', 'TORCH_DIR=' + os.path.dirname(torch.__file__)],
                         capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-8000:])
    if res.returncode != 0:
        raise RuntimeError('building libhcmoco_hip.so failed (see output above)')
    return LIB_PATH


def lib():
    """The loaded library with argtypes set.  Raises if it has not been built -- there is no
    fallback implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'libhcmoco_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ '
                'as g; g.build()"` or `make -C hcmoco_amd/csrc`. hcmoco_amd has no CPU/eager fallback.'
                % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError: the library does not match the header
            fn.restype = res
            fn.argtypes = args
        if handle.hcm_abi_version() != ABI_VERSION:
            raise ImportError('libhcmoco_hip.so has ABI version %d, this package binds version %d: rebuild it '
                              '(make -C hcmoco_amd/csrc)' % (handle.hcm_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


_glue_loaded = False


def torch_glue():
    """Load csrc/libhcmoco_torch.so (registers torch.ops.hcmoco.*: C++ autograd nodes over the C ABI).
    Raises if it has not been built -- there is no fallback implementation."""
    global _glue_loaded
    if not _glue_loaded:
        lib()
        if not os.path.exists(GLUE_PATH):
            raise ImportError('libhcmoco_torch.so is missing (%s); build it with `make -C hcmoco_amd/csrc`' % GLUE_PATH)
        import torch
        torch.ops.load_library(GLUE_PATH)
        _glue_loaded = True
    import torch
    return torch.ops.hcmoco


def check(rc, what):
    if rc != 0:
        msg = lib().hcm_error_string(rc)
        raise HipError('%s failed: hip error %d (%s)' % (what, rc, msg.decode() if msg else '?'))
