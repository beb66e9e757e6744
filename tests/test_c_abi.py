"""The C-ABI library loads without a GPU and exports exactly what include/hcmoco_hip.h declares
(no compute calls here).  Also: host-only entry points, argument validation, loud failure paths."""
import ctypes as C
import os
import re

import pytest
import torch

from hcmoco_amd import _lib, hip_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'hcmoco_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hcm_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 30
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes table binds the same set, no more, no less
    assert sorted(_lib.SIGNATURES) == names


def test_every_entry_point_cites_the_reference_interface_it_replaces():
    src = open(os.path.join(ROOT, 'include', 'hcmoco_hip.h')).read()
    for needle in ('memory/alias_multinomial.py', 'memory/mem_bank.py', 'learning/contrast_trainer.py',
                   'memory/mem_moco.py', 'src/pointnet2_api.cpp', 'src/ball_query_gpu.h', 'src/sampling_gpu.h',
                   'src/group_points_gpu.h', 'src/interpolate_gpu.h'):
        assert needle in src, needle


def test_abi_version_and_error_string():
    L = _lib.lib()
    assert L.hcm_abi_version() == _lib.ABI_VERSION == 6
    assert re.search(r'#define HCM_ABI_VERSION %d\b' % _lib.ABI_VERSION, open(_lib.HEADER).read())
    assert b'invalid' in L.hcm_error_string(1).lower()


def test_host_alias_build_matches_reference_tables(golden):
    g = golden('alias_tables')
    prob, alias = hip_ops.alias_build(g['probs'])
    assert torch.equal(prob, g['prob']) and torch.equal(alias, g['alias'])
    prob, alias = hip_ops.alias_build(torch.ones(1000))
    assert torch.equal(prob, g['uni1000_prob']) and torch.equal(alias, g['uni1000_alias'])


def test_workspace_sizes_are_consistent_and_grow_with_the_problem():
    L = _lib.lib()
    a = L.hcm_bank_nce_workspace_bytes(32, 16385, 128)
    b = L.hcm_bank_nce_workspace_bytes(32, 65537, 128)
    assert 0 < a < b and a % 16 == 0
    assert L.hcm_dense_soft_nce_workspace_bytes(32, 400, 128) > 2 * 2 * 32 * 400 * 128 * 4
    assert L.hcm_joint_nce_workspace_bytes(32, 17, 128) > 0 and L.hcm_scl_workspace_bytes(32, 17, 128) > 0


def test_argument_validation_returns_errors_not_crashes():
    L = _lib.lib()
    z = C.c_void_p(0)
    assert L.hcm_bank_nce_fused(z, z, z, 10, z, z, z, z, z, z, 4, 8, 96, 0.07, z, z, z, z, z, z, 0, z) != 0   # D=96
    assert L.hcm_bank_update(z, z, z, 10, z, z, z, z, 0, 128, 0.5, z) != 0                                      # BW=0
    assert L.hcm_joint_nce(z, z, _lib.Strides4(1, 1, 1, 1), 2, 128, 8, 8, z, z, z, z, 40, 0.07,
                           z, z, z, z, z, 0, z) != 0                                                           # J>32
    with pytest.raises(_lib.HipError):
        _lib.check(1, 'demo')


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip_ops.bank_update([torch.zeros(4, 128)] * 3, [torch.zeros(2, 128)] * 3, torch.zeros(2, dtype=torch.long), 0.5)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip_ops.fmap_losses(torch.zeros(1, 128, 4, 4), torch.zeros(1, 128, 4, 4), None, None, None, None, None,
                            None, None, 0.07)
    import hcmoco_amd.pointnet2_hip as pn
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pn.three_nn_wrapper(1, 2, 2, torch.zeros(1, 2, 3), torch.zeros(1, 2, 3), torch.zeros(1, 2, 3),
                            torch.zeros(1, 2, 3, dtype=torch.int32))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under hcmoco_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'hcmoco_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or '/root/reference' in txt.replace(
                        '/root/reference/pycontrast', '').replace('(/root/reference', ''):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
