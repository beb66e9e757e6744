"""Encoder plugin surface vs the reference (fixtures from tests/golden/gen_golden.py): identical
state_dict keys/shapes/parameter count, and -- with name-keyed deterministic weights that need no
reference code to reproduce -- identical forward outputs in eval and train mode.  CPU only."""
import argparse
import zlib

import numpy as np
import pytest
import torch

from hcmoco_amd.pycontrast.networks.build_backbone import build_model


def deterministic_fill(state_dict):
    out = {}
    for k, v in state_dict.items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
        elif k.endswith('running_var'):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith('running_mean'):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        elif v.dim() == 1 and k.endswith('.weight'):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith('.bias'):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        elif k.endswith('.e'):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            fan_in = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 1
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / np.sqrt(fan_in))
    return out


def make_opt(skel):
    return argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                              in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                              skeleton_meta_name=skel, IN_Pretrain=None, depth_Pretrain=None, mem='bank')


@pytest.mark.parametrize('skel', ['mpii', 'coco_reduce'])
def test_state_dict_and_forward_match_reference(golden, skel):
    g = golden('model_hrnet_w18_' + skel)
    model, ema = build_model(make_opt(skel))
    assert ema is None
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['keys']]
    assert [str(list(v.shape)) for v in sd.values()] == [str(s) for s in g['shapes']]
    assert sum(p.numel() for p in model.parameters()) == g['n_params']
    model.load_state_dict(deterministic_fill(sd))
    for mode in ('eval', 'train'):
        getattr(model, mode)()
        with torch.no_grad():
            f1, f2, f3, f, aux = model(g['x'], g['s'], return_fm=True)
        tol = dict(rtol=1e-4, atol=1e-5)
        assert torch.allclose(f, g[mode + '_f'], **tol)
        assert torch.allclose(f3, g[mode + '_feat3'], **tol)
        assert torch.allclose(aux['linear_merge1'][:, :8, ::5, ::5], g[mode + '_lm1_slice'], **tol)
        assert torch.allclose(aux['linear_merge2'][:, :8, ::5, ::5], g[mode + '_lm2_slice'], **tol)
        assert torch.allclose(f1[3], g[mode + '_feat1_3'], **tol)
        assert torch.allclose(f2[0][:, :, ::7, ::7], g[mode + '_feat2_0_slice'], **tol)
        assert [tuple(m.shape[1:]) for m in f1] == [(18, 16, 16), (36, 8, 8), (72, 4, 4), (144, 2, 2)]
        assert f.shape == (2, 384) and aux['linear_merge1'].shape == (2, 128, 16, 16)


def test_w18_parameter_count_and_plain_forward():
    model, _ = build_model(make_opt('mpii'))
    assert sum(p.numel() for p in model.parameters()) == 19579252          # SURVEY 2.4 [probed]
    f = model(torch.randn(2, 6, 64, 64), torch.rand(2, 16, 2))
    assert f.shape == (2, 384)
    n = f.view(2, 3, 128).norm(dim=2)
    assert torch.allclose(n, torch.ones_like(n), atol=1e-5)                # three L2-normalised heads


def test_coco17_extension_and_moco_ema():
    opt = make_opt('coco17')
    opt.mem = 'moco'
    model, ema = build_model(opt)
    assert ema is not None
    assert model(torch.randn(1, 6, 64, 64), torch.rand(1, 17, 2), mode=2).shape == (1, 270 * 2 + 128)


def test_batch_counters_advance_once_per_training_forward():
    model, _ = build_model(make_opt('mpii'))
    model.train()
    model(torch.randn(2, 6, 64, 64), torch.rand(2, 16, 2))
    model(torch.randn(2, 6, 64, 64), torch.rand(2, 16, 2))
    sd = model.state_dict()
    counters = [v for k, v in sd.items() if k.endswith('num_batches_tracked') and k.startswith('encoder1')]
    assert len(counters) > 300 and all(int(c) == 2 for c in counters)       # as nn.BatchNorm2d would count
    model.eval()
    model(torch.randn(1, 6, 64, 64), torch.rand(1, 16, 2))
    assert int(sd['encoder1.bn1.num_batches_tracked']) == 2
