"""Whole-step checker (TEST INFRASTRUCTURE): re-evaluates the records a
``hcmoco_amd.pycontrast.learning.engine.RecordingEngine`` took of one training step -- the tensors the HIP loss
kernels were given and what they returned -- with the oracle (oracle/hcmoco_oracle.py, a restatement of
/root/reference/pycontrast/learning/contrast_trainer.py:894-1039 and memory/mem_bank.py:172-205) on the CPU.

Used by tests/test_whole_step_gpu.py (one step per BASELINE config at the config's own size) and by the
``--check`` leg of bench.py, which runs ``python -m oracle.check_step <records.pt>`` in a CPU-only subprocess
(the bench process itself never imports the oracle).  Only tests/, smoke() and bench.py's checker/baseline legs
may use this module; the product never does."""
import json
import sys

import torch
import torch.nn.functional as F

from oracle import hcmoco_oracle as O

FP32 = {'loss_rtol': 1e-5, 'loss_atol': 1e-6, 'grad_rel_l2': 1e-4, 'meter_rtol': 1e-5, 'meter_atol': 1e-6,
        'fmap_grad_rel_l2': 1e-4, 'update_atol': 1e-6}
# bf16 feature-map contractions / bf16 bank storage (BASELINE config 5): tolerances restated against the fp32 oracle
BF16_FMAP = {'meter_rtol': 1e-2, 'meter_atol': 1e-3, 'fmap_grad_rel_l2': 2e-2}


def rel_l2(a, b):
    a, b = a.double(), b.double()
    n = float(b.norm())
    return float((a - b).norm()) / n if n > 0 else float(a.abs().max())


def check_bank(rec, tol):
    """rows 1-4: idx[:,0] == index (bit-exact), six losses / accuracies, d/dx of all B rows per modality, the
    momentum update of the touched rows (last duplicate wins) and bit-identity of the untouched ones."""
    rep = {}
    bf16 = rec['banks0'][0].dtype == torch.bfloat16
    banks = [b.float() for b in rec['banks0']]            # bf16 banks: the oracle reads the same rounded rows
    idx, xs = rec['idx'], [x.float() for x in rec['x']]
    assert torch.equal(idx[:, 0], rec['index'].clamp(0, banks[0].shape[0] - 1)), 'idx[:,0] != index'
    assert int(idx.min()) >= 0 and int(idx.max()) < banks[0].shape[0]
    lo, ao, go = O.bank_nce_chunked(banks, idx, xs, rec['T'], rec['use_depth'], rec['use_rgb'])
    l, a = rec['losses'].double(), rec['accs'].double()
    assert torch.allclose(l, lo, rtol=tol['loss_rtol'], atol=tol['loss_atol']), ('bank losses', l, lo)
    assert torch.allclose(a, ao, atol=1e-3), ('bank accs', a, ao)
    rep['bank_loss_max_rel'] = float(((l - lo).abs() / lo.abs().clamp_min(1e-12)).max())
    assert abs(float(rec['total']) - float(lo.sum())) <= tol['loss_rtol'] * 6 * abs(float(lo.sum())) + 1e-6
    errs = []
    for i in range(3):
        g = rec['grads'].get('x%d' % (i + 1))
        assert g is not None, 'no gradient recorded for x%d' % (i + 1)
        if float(go[i].norm()) == 0:
            assert float(g.abs().max()) == 0
            continue
        errs.append(rel_l2(g, go[i]))
        assert errs[-1] < tol['grad_rel_l2'], ('bank grad', i, errs[-1])
    rep['bank_grad_max_rel_l2'] = max(errs) if errs else 0.0
    assert all(rec['untouched_rows_unchanged']), 'bank rows outside all_index changed'
    upd = []
    for i in range(3):
        ref = O.bank_update(banks[i], rec['all_x'][i].float(), rec['all_index'], rec['m']).index_select(0, rec['all_index'])
        err = (rec['after_rows'][i].float() - ref).abs()
        if bf16:
            assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all()), ('bank update (bf16)', i, float(err.max()))
        else:
            assert float(err.max()) <= tol['update_atol'], ('bank update', i, float(err.max()))
        upd.append(float(err.max()))
    rep['bank_update_max_abs'] = max(upd)
    return rep


def _fmap_oracle(map1, map2, rec):
    keepb = rec['keep'].bool()
    ud = rec['use_depth']
    ld, ad, d1, d2 = O.dense_soft_nce(map1, map2, rec['sample_ind'][keepb], keepb, rec['temperature'],
                                      None if bool(keepb.any()) else torch.zeros(1))
    lj, aj, j1, j2, j3 = O.joint_nce(map1, map2, rec['feat3'].float(), rec['joints2d'], rec['joints_vis'],
                                     rec['temperature'], ud)
    udv = ud if ud is not None else torch.ones(map1.shape[0])
    ls, s1, s2, _ = O.scl(map1, map2, rec['joints2d'], rec['temperature'], udv, rec['use_rgb'])
    meters = torch.cat([ld, ad, lj, aj, ls.reshape(1)])
    return meters, d1 + j1 + s1, d2 + j2 + s2, j3


def _cmp_meters(rec, want, tol, rep):
    got = rec['meters'].double()
    want = want.double()
    nan = torch.isnan(want)
    assert torch.equal(torch.isnan(got), nan), ('fmap meters NaN pattern', got, want)
    assert torch.allclose(got[~nan], want[~nan], rtol=tol['meter_rtol'], atol=tol['meter_atol']), ('fmap meters', got, want)
    rep['fmap_meter_max_rel'] = float(((got[~nan] - want[~nan]).abs() / want[~nan].abs().clamp_min(1e-6)).max())
    tot = want[[0, 1, 4, 5, 8]].sum()
    if not bool(torch.isnan(tot)):
        assert abs(float(rec['total']) - float(tot)) <= 5 * tol['meter_rtol'] * abs(float(tot)) + 5 * tol['meter_atol']


def _cmp_grads(rec, want, tol, rep):
    errs = {}
    for k, ref in want.items():
        g = rec['grads'].get(k)
        assert g is not None, 'no gradient recorded for %s' % k
        if bool(torch.isnan(ref).any()):
            continue
        if float(ref.norm()) == 0:
            assert float(g.abs().max()) == 0, k
            continue
        errs[k] = rel_l2(g, ref)
        assert errs[k] < tol['fmap_grad_rel_l2'], ('fmap grad', k, errs[k])
    rep['fmap_grad_max_rel_l2'] = max(errs.values()) if errs else 0.0


def check_fmap(rec, tol):
    """rows 5-7 on full maps: the nine meters and d/d(map1, map2, feat3)."""
    rep = {}
    want, g1, g2, g3 = _fmap_oracle(rec['map1'].float(), rec['map2'].float(), rec)
    _cmp_meters(rec, want, tol, rep)
    _cmp_grads(rec, {'map1': g1, 'map2': g2, 'feat3': g3}, tol, rep)
    return rep


def check_fmap_sampled(rec, tol):
    """rows 5-8 from the raw HRNet branch maps: the reference data flow (merge_all_res = 3 bilinear up-samplings +
    concat, the full-resolution 1x1 projection, build_backbone.py:243-254, :290-300) in plain torch on the CPU, the
    oracle losses on the full maps, and their gradients pulled back through that flow by torch autograd to the
    branch maps and the projection weights."""
    rep = {}

    def leaf(t):
        return t.float().clone().requires_grad_(True)

    def project(branches, w, b):
        size = branches[0].shape[-2:]
        up = [branches[0]] + [F.interpolate(m, size=size, mode='bilinear', align_corners=False) for m in branches[1:]]
        return F.conv2d(torch.cat(up, 1), w, b)

    b1, b2 = [leaf(t) for t in rec['branches1']], [leaf(t) for t in rec['branches2']]
    w1, c1 = leaf(rec['proj1'][0]), leaf(rec['proj1'][1])
    w2, c2 = leaf(rec['proj2'][0]), leaf(rec['proj2'][1])
    map1, map2 = project(b1, w1, c1), project(b2, w2, c2)
    want, g1, g2, g3 = _fmap_oracle(map1.detach(), map2.detach(), rec)
    _cmp_meters(rec, want, tol, rep)
    torch.autograd.backward([map1, map2], [g1, g2])
    ref = {'feat3': g3, 'proj1_w': w1.grad, 'proj1_b': c1.grad, 'proj2_w': w2.grad, 'proj2_b': c2.grad}
    ref.update({'b1_%d' % i: t.grad for i, t in enumerate(b1)})
    ref.update({'b2_%d' % i: t.grad for i, t in enumerate(b2)})
    _cmp_grads(rec, ref, tol, rep)
    return rep


def check_section(rec, tol):
    """The fused loss section (engine.section, csrc/section.hip): rows 1-9 of SURVEY 8a in one record.  Oracle flow,
    all on the CPU: pooling + heads (torch autograd over O.heads) -> f; bank NCE / update on f; merge_all_res + full
    1x1 projection (torch autograd) -> oracle feature-map losses; the oracle's closed-form gradients w.r.t. f and the
    two projected maps are pulled back by torch autograd to the eight branch maps, the SemGCN output, the three
    heads and the two projections.  Everything the HIP section returned is compared: f, idx[:,0], the sampled pixels'
    validity, 6 + 6 + 9 meters, the bank update and all 19 gradients."""
    rep = {}

    def leaf(t):
        return t.float().clone().requires_grad_(True)

    b1, b2 = [leaf(t) for t in rec['branches1']], [leaf(t) for t in rec['branches2']]
    feat3 = leaf(rec['feat3'])
    hw = [leaf(w) for w, _ in rec['heads']]
    hb = [leaf(b) for _, b in rec['heads']]
    f = O.heads(b1, b2, feat3, hw, hb)
    F_ = f.shape[1] // 3
    assert torch.allclose(rec['f'].float(), f.detach(), rtol=1e-5, atol=1e-6), ('heads f', float((rec['f'] - f.detach()).abs().max()))
    rep['f_max_abs'] = float((rec['f'].float() - f.detach()).abs().max())
    # ---- bank on the recorded f (the section's own features: errors do not compound)
    fr = rec['f'].float()
    bank = dict(rec)
    bank.update(x=[fr[:, i * F_:(i + 1) * F_].contiguous() for i in range(3)],
                use_rgb=None if rec['stage2'] else rec['use_rgb'], grads={})
    bf16 = rec['banks0'][0].dtype == torch.bfloat16
    banks = [b.float() for b in rec['banks0']]
    idx = rec['idx']
    assert torch.equal(idx[:, 0], rec['index'].clamp(0, banks[0].shape[0] - 1)), 'idx[:,0] != index'
    lo, ao, go = O.bank_nce_chunked(banks, idx, bank['x'], rec['T'], rec['use_depth'], bank['use_rgb'])
    l, a = rec['losses'].double(), rec['accs'].double()
    assert torch.allclose(l, lo, rtol=tol['loss_rtol'], atol=tol['loss_atol']), ('bank losses', l, lo)
    assert torch.allclose(a, ao, atol=1e-3), ('bank accs', a, ao)
    rep['bank_loss_max_rel'] = float(((l - lo).abs() / lo.abs().clamp_min(1e-12)).max())
    assert all(rec['untouched_rows_unchanged']), 'bank rows outside all_index changed'
    upd = []
    for i in range(3):
        ref = O.bank_update(banks[i], rec['all_x'][i].float(), rec['all_index'], rec['m']).index_select(0, rec['all_index'])
        err = (rec['after_rows'][i].float() - ref).abs()
        if bf16:
            assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all()), ('bank update (bf16)', i, float(err.max()))
        else:
            assert float(err.max()) <= tol['update_atol'], ('bank update', i, float(err.max()))
        upd.append(float(err.max()))
    rep['bank_update_max_abs'] = max(upd)
    gf = torch.cat([g.float() for g in go], 1)
    total = float(lo.sum())
    if rec['stage2']:
        # ---- sampled pixels must be valid draws: inside the resized mask of kept images
        B = rec['index'].shape[0]
        h, w = rec['branches1'][0].shape[-2:]
        keep_ref, m = O.dense_keep(rec['depth_mask'], h, w)
        ud = rec['use_depth']
        if ud is not None:
            keep_ref = keep_ref & bool(ud.sum() > 0)
        assert torch.equal(rec['keep'].bool(), keep_ref), 'keep flags'
        si = rec['sample_ind']
        assert bool((m.gather(1, si)[keep_ref] > 0).all()), 'a sampled pixel lies outside the depth mask'
        assert torch.equal(rec['pix'][:, si.shape[1]:], O.joint_pixels(rec['joints2d'], h)), 'joint pixels'

        def project(branches, w, b):
            size = branches[0].shape[-2:]
            up = [branches[0]] + [F.interpolate(x, size=size, mode='bilinear', align_corners=False) for x in branches[1:]]
            return F.conv2d(torch.cat(up, 1), w, b)

        pw = [leaf(w_) for w_, _ in rec['projs']]
        pb = [leaf(b_) for _, b_ in rec['projs']]
        map1, map2 = project(b1, pw[0], pb[0]), project(b2, pw[1], pb[1])
        frec = dict(rec, feat3=rec['feat3'])
        want, g1, g2, g3 = _fmap_oracle(map1.detach(), map2.detach(), frec)
        fm = dict(rec)
        fm['total'] = rec['total'] - float(rec['losses'].sum())
        _cmp_meters(fm, want, tol, rep)
        torch.autograd.backward([f, map1, map2, feat3], [gf, g1, g2, g3])
    else:
        torch.autograd.backward([f], [gf])
    ref = {'feat3': feat3.grad}
    ref.update({'b1_%d' % i: t.grad for i, t in enumerate(b1)})
    ref.update({'b2_%d' % i: t.grad for i, t in enumerate(b2)})
    for i in range(3):
        ref['head%d_w' % (i + 1)] = hw[i].grad
        ref['head%d_b' % (i + 1)] = hb[i].grad
    if rec['stage2']:
        for i in range(2):
            ref['proj%d_w' % (i + 1)] = pw[i].grad
            ref['proj%d_b' % (i + 1)] = pb[i].grad
    _cmp_grads(rec, ref, tol, rep)
    return rep


def check_section_pn(rec, tol):
    """The fused loss section of the HRNetPN model (hip_ops.stage2_section_pn; reference: networks/build_backbone.py:457-514,
    learning/contrast_trainer.py:894-1039).  Same oracle flow as ``check_section``; the second modality is the point-cloud
    encoder: head 2 reads the mean over the points of ``feat2 [B, C2, Npts]`` (:486) and the depth feature map
    ``lm2 [B, F, h, w]`` enters the losses as it is (the model produced it: Conv1d + pts2depth + nearest resize, :499-505).
    Compared: f, idx[:, 0], sampled pixels, 6 + 6 + 9 meters, the bank update, and the gradients of the four HRNet branch
    maps, feat2, lm2, the SemGCN output, the three heads and encoder1_linear."""
    rep = {}

    def leaf(t):
        return t.float().clone().requires_grad_(True)

    b1 = [leaf(t) for t in rec['branches1']]
    feat2, lm2 = leaf(rec['branches2'][0]), leaf(rec['branches2'][1])
    feat3 = leaf(rec['feat3'])
    hw = [leaf(w) for w, _ in rec['heads']]
    hb = [leaf(b) for _, b in rec['heads']]
    x = [torch.cat([m.mean(dim=(2, 3)) for m in b1], 1), feat2.mean(-1), feat3.mean(1)]
    f = torch.cat([F.normalize(F.linear(xi, Wi, bi), p=2, dim=1) for xi, Wi, bi in zip(x, hw, hb)], 1)
    F_ = f.shape[1] // 3
    assert torch.allclose(rec['f'].float(), f.detach(), rtol=1e-5, atol=1e-6), ('heads f', float((rec['f'] - f.detach()).abs().max()))
    rep['f_max_abs'] = float((rec['f'].float() - f.detach()).abs().max())
    fr = rec['f'].float()
    xs = [fr[:, i * F_:(i + 1) * F_].contiguous() for i in range(3)]
    banks = [b.float() for b in rec['banks0']]
    idx = rec['idx']
    assert torch.equal(idx[:, 0], rec['index'].clamp(0, banks[0].shape[0] - 1)), 'idx[:,0] != index'
    lo, ao, go = O.bank_nce_chunked(banks, idx, xs, rec['T'], rec['use_depth'], None)
    l, a = rec['losses'].double(), rec['accs'].double()
    assert torch.allclose(l, lo, rtol=tol['loss_rtol'], atol=tol['loss_atol']), ('bank losses', l, lo)
    assert torch.allclose(a, ao, atol=1e-3), ('bank accs', a, ao)
    rep['bank_loss_max_rel'] = float(((l - lo).abs() / lo.abs().clamp_min(1e-12)).max())
    assert all(rec['untouched_rows_unchanged']), 'bank rows outside all_index changed'
    upd = []
    for i in range(3):
        ref = O.bank_update(banks[i], rec['all_x'][i].float(), rec['all_index'], rec['m']).index_select(0, rec['all_index'])
        err = (rec['after_rows'][i].float() - ref).abs()
        assert float(err.max()) <= tol['update_atol'], ('bank update', i, float(err.max()))
        upd.append(float(err.max()))
    rep['bank_update_max_abs'] = max(upd)
    gf = torch.cat([g.float() for g in go], 1)
    h, w = rec['branches1'][0].shape[-2:]
    keep_ref, m = O.dense_keep(rec['depth_mask'], h, w)
    ud = rec['use_depth']
    if ud is not None:
        keep_ref = keep_ref & bool(ud.sum() > 0)
    assert torch.equal(rec['keep'].bool(), keep_ref), 'keep flags'
    si = rec['sample_ind']
    assert bool((m.gather(1, si)[keep_ref] > 0).all()), 'a sampled pixel lies outside the depth mask'
    assert torch.equal(rec['pix'][:, si.shape[1]:], O.joint_pixels(rec['joints2d'], h)), 'joint pixels'
    pw, pb = leaf(rec['projs'][0][0]), leaf(rec['projs'][0][1])
    up = [b1[0]] + [F.interpolate(t, size=(h, w), mode='bilinear', align_corners=False) for t in b1[1:]]
    map1 = F.conv2d(torch.cat(up, 1), pw, pb)
    want, g1, g2, g3 = _fmap_oracle(map1.detach(), lm2.detach(), rec)
    fm = dict(rec)
    fm['total'] = rec['total'] - float(rec['losses'].sum())
    _cmp_meters(fm, want, tol, rep)
    torch.autograd.backward([f, map1, lm2, feat3], [gf, g1, g2, g3])
    ref = {'feat3': feat3.grad, 'feat2': feat2.grad, 'lm2': lm2.grad, 'proj1_w': pw.grad, 'proj1_b': pb.grad}
    ref.update({'b1_%d' % i: t.grad for i, t in enumerate(b1)})
    for i in range(3):
        ref['head%d_w' % (i + 1)] = hw[i].grad
        ref['head%d_b' % (i + 1)] = hb[i].grad
    _cmp_grads(rec, ref, tol, rep)
    return rep


def check_records(records, tol=None):
    """-> report dict; raises AssertionError on the first disagreement."""
    report = {'calls': {}}
    for rec in records:
        t = dict(FP32)
        if rec.get('fmap_dtype') == 'bf16':
            t.update(BF16_FMAP)
        if tol:
            t.update(tol)
        fn = {'bank': check_bank, 'fmap': check_fmap, 'fmap_sampled': check_fmap_sampled, 'section': check_section,
              'section_pn': check_section_pn}[rec['kind']]
        rep = fn(rec, t)
        report['calls'][rec['kind']] = report['calls'].get(rec['kind'], 0) + 1
        for k, v in rep.items():
            report[k] = max(report.get(k, 0.0), v)
    return report


def main(argv):
    """``python -m oracle.check_step records.pt`` -> one JSON line {"checked": bool, ...}."""
    records = torch.load(argv[0], map_location='cpu', weights_only=False)
    try:
        rep = check_records(records)
        rep['checked'] = True
    except AssertionError as e:
        rep = {'checked': False, 'error': str(e)[:500]}
    print(json.dumps(rep))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
