"""Plumbing of the whole-step checker on the CPU: a ``RecordingEngine`` around the oracle-backed engine records one
stage-2 step of the real trainer loop (both data flows: full maps and branch maps + projection), the records go
through ``torch.save`` like in bench.py, and ``oracle/check_step.py`` must accept them -- and must REJECT them once
a recorded number is off by more than its tolerance.  (What the records are checked against on the GPU box is the
HIP path; here the point is that recorder, file format and checker agree and that the checker has teeth.)"""
import copy
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

from hcmoco_amd.pycontrast import main_contrast
from hcmoco_amd.pycontrast.learning.engine import RecordingEngine
from oracle.check_step import check_records
from oracle.oracle_engine import OracleLossEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(sampled, tmp):
    from conftest import free_port
    os.environ['MASTER_PORT'] = str(free_port())
    argv = ['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18',
            '--in_channel_list', '3,3', '--batch_size', '4', '--nce_k', '64', '--world-size', '1', '--dist-backend', 'gloo',
            '--synthetic', '--synthetic_n_data', '256', '--synthetic_size', '64', '--synthetic_steps', '1', '--epochs', '1',
            '--print_freq', '1', '--save_freq', '1', '--model_path', tmp, '--tb_path', tmp, '--seed', '3',
            '--learning_rate', '0.01', '--linear_feat_map', '1', '--modality_missing', '1',
            '--pri3d_num_samples_per_image', '16', '--sampled_projection', str(sampled)]
    eng = RecordingEngine(inner=OracleLossEngine())
    try:
        main_contrast.main(argv, engine=eng)
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    return eng.records


@pytest.mark.parametrize('sampled', [1, 0])
def test_recorded_step_passes_and_a_wrong_number_fails(sampled, tmp_path):
    recs = _records(sampled, str(tmp_path))
    assert [r['kind'] for r in recs] == ['bank', 'fmap_sampled' if sampled else 'fmap']
    path = str(tmp_path / 'rec.pt')
    torch.save(recs, path)
    recs = torch.load(path, map_location='cpu', weights_only=False)
    rep = check_records(recs)
    assert rep['calls'] == {'bank': 1, ('fmap_sampled' if sampled else 'fmap'): 1}
    assert rep['bank_grad_max_rel_l2'] < 1e-5 and rep['fmap_grad_max_rel_l2'] < 1e-5
    # teeth: every class of recorded quantity, perturbed beyond its tolerance, is caught
    def broken(mutate):
        bad = copy.deepcopy(recs)
        mutate(bad)
        with pytest.raises(AssertionError):
            check_records(bad)
    broken(lambda r: r[0]['losses'].__setitem__(2, r[0]['losses'][2] * (1 + 1e-4)))
    broken(lambda r: r[0]['accs'].__setitem__(0, r[0]['accs'][0] + 25.0))
    broken(lambda r: r[0]['grads']['x2'].mul_(1.001))
    broken(lambda r: r[0]['after_rows'][1].__setitem__((0, 5), r[0]['after_rows'][1][0, 5] + 1e-4))
    broken(lambda r: r[0]['idx'].__setitem__((1, 0), (r[0]['idx'][1, 0] + 1) % 256))
    broken(lambda r: r[0].__setitem__('untouched_rows_unchanged', [True, False, True]))
    broken(lambda r: r[1]['meters'].__setitem__(0, r[1]['meters'][0] * 1.01))
    broken(lambda r: r[1]['grads']['feat3'].mul_(1.01))
    key = 'b2_3' if sampled else 'map2'
    broken(lambda r: r[1]['grads'][key].mul_(1.01))
    if sampled:
        broken(lambda r: r[1]['grads']['proj1_w'].add_(r[1]['grads']['proj1_w'].abs().mean() * 0.01))


def test_checker_command_line(tmp_path):
    """bench.py --check runs the checker as ``python -m oracle.check_step file`` in a CPU-only process."""
    recs = _records(1, str(tmp_path))
    path = str(tmp_path / 'rec.pt')
    torch.save(recs, path)
    env = dict(os.environ, PYTHONPATH=ROOT, HIP_VISIBLE_DEVICES='', OMP_NUM_THREADS='4')
    res = subprocess.run([sys.executable, '-m', 'oracle.check_step', path], capture_output=True, text=True, env=env,
                         cwd=ROOT, timeout=600)
    rep = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert rep['checked'] is True and rep['calls'] == {'bank': 1, 'fmap_sampled': 1}, (rep, res.stderr[-500:])
    recs[0]['losses'][0] += 1.0
    torch.save(recs, path)
    res = subprocess.run([sys.executable, '-m', 'oracle.check_step', path], capture_output=True, text=True, env=env,
                         cwd=ROOT, timeout=600)
    rep = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert rep['checked'] is False and 'bank losses' in rep['error']


def test_oracle_heads_equal_the_model_heads():
    """O.heads (what the whole-step checker derives f from) against the model's own pooling + Linear + Normalize
    (networks/build_backbone.py:265-288; the model itself is pinned to the reference in tests/test_model_surface.py)."""
    import argparse
    from hcmoco_amd.pycontrast.networks.build_backbone import build_model
    from oracle import hcmoco_oracle as O
    torch.manual_seed(1)
    opt = argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                             in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                             skeleton_meta_name='mpii', IN_Pretrain=None, depth_Pretrain=None, mem='bank')
    model, _ = build_model(opt)
    model.train()
    x, s = torch.randn(2, 6, 64, 64), torch.rand(2, 16, 2) * 2 - 1
    f1, f2, f3, f, _ = model(x, s, return_fm=True)
    heads = [model.head1[0], model.head2[0], model.head3[0]]
    ref = O.heads(f1, f2, f3, [l.weight for l in heads], [l.bias for l in heads])
    assert torch.allclose(f, ref, rtol=1e-6, atol=1e-7)
    # with the trainer's flags the model hands back the raw maps and no f
    model.defer_projection = model.defer_heads = True
    out = model(x, s, return_fm=True)
    assert out[3] is None and out[4]['linear_merge1'] is None and len(out[0]) == 4


def test_oracle_pixel_sampler_properties():
    """O.pixel_sample_philox (bit-exact model of hcm_pixel_sample): valid pixels only, dropped images, the
    use_depth early-out, determinism in (seed, offset), uniformity."""
    from oracle import hcmoco_oracle as O
    torch.manual_seed(0)
    B, H, h, S = 4, 64, 16, 50
    mask = (torch.rand(B, H, H) < 0.2).float()
    mask[2] = 0
    ind, keep = O.pixel_sample_philox(mask, h, h, S, None, 7, 1)
    m = O.nearest_resize_mask(mask, h, h).reshape(B, -1)
    assert keep.tolist() == [True, True, False, True] and ind.shape == (B, S)
    assert bool((m.gather(1, ind)[keep] > 0).all()) and bool((ind[2] == 0).all())
    ind2, _ = O.pixel_sample_philox(mask, h, h, S, None, 7, 1)
    ind3, _ = O.pixel_sample_philox(mask, h, h, S, None, 7, 2)
    assert torch.equal(ind, ind2) and not torch.equal(ind, ind3)
    _, keep0 = O.pixel_sample_philox(mask, h, h, S, torch.zeros(B), 7, 1)
    assert not bool(keep0.any())
    big, _ = O.pixel_sample_philox(mask[:1], h, h, 100000, None, 3, 4)
    valid = m[0] > 0
    p = torch.bincount(big[0], minlength=h * h).double() / 100000
    assert float(p[~valid].sum()) == 0 and float((p[valid] - 1.0 / int(valid.sum())).abs().max()) < 0.01
