"""SURVEY 8c "harness rows": two consecutive steps of the reference's OWN training loops
(learning/contrast_trainer.py:532-640 stage 1, :894-1039 stage 2), recorded by
tests/golden/gen_golden.py:gen_trace with the reference CMCMem3, torch SGD and the stand-in encoder of
tests/golden/standin.py.  This repo's trainer replays them from the same initial weights / banks /
batches with the recorded random draws injected (negative indices, sampled pixels):

  * per-step losses and accuracies of every loss term,
  * the banks after each step (checksum + every row): step 2 gathers rows step 1 wrote, the update
    happens after the reads, the duplicate index inside a batch resolves to its last occurrence,
  * the encoder weights after each SGD step (i.e. the gradients of every loss w.r.t. maps / features).

CPU: oracle engine (pins the oracle + host loop to the sequence).  GPU: the HIP engine, same fixture.
"""
import argparse
import importlib.util
import os

import pytest
import torch

from conftest import GOLDEN, load_golden

_spec = importlib.util.spec_from_file_location('standin', os.path.join(GOLDEN, 'standin.py'))
standin = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(standin)


class ReplayEngine(object):
    """Feeds the recorded draws of step t into the wrapped engine (parity mode of both engines)."""

    def __init__(self, base, fx, stage):
        self.base, self.fx, self.stage, self.t = base, fx, stage, 0
        self.records = []

    def bank(self, contrast, f1, f2, f3, index, *rest, **kw):
        idx = self.fx['s%d_idx' % self.t].clone().to(index.device)
        idx[:, 0] = index
        out = self.base.bank(contrast, f1, f2, f3, index, *rest, idx=idx, **kw)
        self.records.append({'bank_losses': out[1].detach().cpu(), 'bank_accs': out[2].detach().cpu()})
        if self.stage == 1:
            self.t += 1
        return out

    def fmap(self, map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb, num_samples,
             temperature, sample_ind=None, keep=None):
        B, _, h, w = map1.shape
        m = torch.nn.functional.interpolate(depth_mask.unsqueeze(1).float(), size=(h, w), mode='nearest').reshape(B, -1)
        keep = (m.sum(-1) > 0)
        rec = self.fx['s%d_sample_ind' % self.t].to(map1.device)          # [B', S] for the kept images
        ind = torch.zeros(B, num_samples, dtype=torch.long, device=map1.device)
        ind[keep] = rec
        out = self.base.fmap(map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb, num_samples,
                             temperature, sample_ind=ind, keep=keep.to(torch.int32))
        self.records[-1]['fmap'] = out[1].detach().cpu()
        self.t += 1
        return out


def _replay(stage, engine, device):
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    fx = load_golden('trace_stage%d' % stage)
    B, n, K, S, steps = fx['B'], fx['n'], fx['K'], fx['S'], fx['steps']
    model = standin.StandInEncoder()
    model.load_state_dict({k[3:]: v for k, v in fx.items() if k.startswith('w0_')})
    model.to(device).train()
    mem = CMCMem3(128, n, K, fx['T'], fx['m'], seed=1)
    for i, b in enumerate(mem.banks()):
        b.copy_(fx['bank0_%d' % (i + 1)])
    mem.to(device)
    opt = torch.optim.SGD(model.parameters(), lr=fx['lr'], momentum=fx['momentum'], weight_decay=fx['weight_decay'])
    args = argparse.Namespace(modality_missing=1, arch='HRNet', pri3d_num_samples_per_image=S, temperature=fx['T'],
                              amp=False, mem='bank' if stage == 1 else 'bank+jointspri3d', rank=0, local_rank=0)
    eng = ReplayEngine(engine, fx, stage)
    tr = ContrastTrainer(args, engine=eng)
    tr.device = torch.device(device)
    after = []
    for t in range(steps):
        batch = [fx['s%d_data%d' % (t, i)] for i in range(12)]
        out = tr.train_step(batch, model, mem, opt, stage2=(stage == 2))
        after.append(([b.detach().float().cpu().clone() for b in mem.banks()],
                      {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                      float(out['loss'])))
    return fx, eng.records, after


def _check(fx, records, after, stage, loss_rtol, bank_atol, w_rtol):
    for t in range(fx['steps']):
        r = records[t]
        want_l, want_a = fx['s%d_bank_losses' % t], fx['s%d_bank_accs' % t]
        assert torch.allclose(r['bank_losses'], want_l, rtol=loss_rtol, atol=1e-6), (t, r['bank_losses'], want_l)
        assert torch.allclose(r['bank_accs'], want_a, atol=1e-3), (t, r['bank_accs'], want_a)
        total = float(want_l.double().sum())
        if stage == 2:
            want = torch.cat([fx['s%d_dense' % t], fx['s%d_joint' % t], fx['s%d_scl' % t]])
            assert torch.allclose(r['fmap'], want, rtol=loss_rtol, atol=1e-5), (t, r['fmap'], want)
            total += float(want[[0, 1, 4, 5, 8]].double().sum())       # losses; the rest are accuracies
        assert abs(after[t][2] - total) <= 2 * loss_rtol * abs(total)
        banks, sd, _ = after[t]
        for i, b in enumerate(banks):
            ref = fx['s%d_bank_%d' % (t, i + 1)]
            touched = (ref != fx['bank0_%d' % (i + 1)]).any(1)
            prev = fx['bank0_%d' % (i + 1)] if t == 0 else fx['s%d_bank_%d' % (t - 1, i + 1)]
            # rows the reference left alone in this step are bit-identical to the previous state
            same = ~(ref != prev).any(1)
            prev_mine = None if t == 0 else after[t - 1][0][i]
            if prev_mine is not None:
                assert torch.equal(b[same], prev_mine[same])
            else:
                assert torch.equal(b[same], prev[same])
            assert bool(touched.any())
            assert (b - ref).abs().max().item() <= bank_atol, (t, i, (b - ref).abs().max().item())
            assert abs(float(b.double().sum()) - fx['s%d_bank_%d_checksum' % (t, i + 1)]) <= 128 * bank_atol
        for k, v in sd.items():
            ref = fx['s%d_w_%s' % (t, k)]
            err = (v - ref).norm().item() / max(ref.norm().item(), 1e-12)
            assert err <= w_rtol, (t, k, err)


@pytest.mark.parametrize('stage', [1, 2])
def test_oracle_replays_reference_training_loop(stage):
    from oracle.oracle_engine import OracleLossEngine
    fx, records, after = _replay(stage, OracleLossEngine(), 'cpu')
    _check(fx, records, after, stage, loss_rtol=1e-5, bank_atol=2e-6, w_rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('stage', [1, 2])
def test_hip_engine_replays_reference_training_loop(stage):
    from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
    fx, records, after = _replay(stage, HipLossEngine(), 'cuda:0')
    _check(fx, records, after, stage, loss_rtol=2e-5, bank_atol=5e-6, w_rtol=1e-4)
