"""Pin the CPU oracle (oracle/hcmoco_oracle.py) to golden vectors produced by the
reference's own code (tests/golden/gen_golden.py).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import hcmoco_oracle as O


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a.double(), b.double(), rtol=rtol, atol=atol), (a - b).abs().max()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_alias_tables(golden):
    g = golden('alias_tables')
    prob, alias = O.alias_build(g['probs'])
    assert torch.equal(prob, g['prob'])
    assert torch.equal(alias, g['alias'])
    for n in (1000, 4096):
        prob, alias = O.alias_build(torch.ones(n))
        assert torch.equal(prob, g['uni%d_prob' % n])
        assert torch.equal(alias, g['uni%d_alias' % n])
        assert bool((prob == 1).all()) and bool((alias == 0).all())


def test_alias_draw_philox_properties():
    prob, alias = O.alias_build(torch.tensor([0.1, 0.2, 0.3, 0.4]))
    d = O.alias_draw_philox(prob, alias, 200000, seed=7, offset=3)
    assert d.min() >= 0 and d.max() < 4
    freq = torch.bincount(d, minlength=4).double() / d.numel()
    assert torch.allclose(freq, torch.tensor([0.1, 0.2, 0.3, 0.4], dtype=torch.double), atol=5e-3)
    # counter-based: prefix property and determinism
    d2 = O.alias_draw_philox(prob, alias, 1000, seed=7, offset=3)
    assert torch.equal(d[:1000], d2)
    # Philox known-answer (Random123 kat: ctr=0,key=0)
    r = O.philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = O.philox4x32_10(np.full((1, 4), 0xffffffff, np.uint32), (0xffffffff, 0xffffffff))[0]
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


REGIMES = ['none', 'depth_mix', 'depth_all0', 'both_mix', 'both_none']


@pytest.mark.parametrize('regime', ['none', 'depth_mix', 'depth_all0', 'both_mix', 'both_none'])
def test_bank_nce_chunked_is_pinned_too(golden, regime):
    """The memory-bounded form the whole-step checker uses at BASELINE sizes (oracle/check_step.py) against the
    reference's own losses / accuracies / gradients."""
    g = golden('bank_nce')
    banks = [g['bank0_%d' % i] for i in (1, 2, 3)]
    xs = [g['x%d' % i] for i in (1, 2, 3)]
    ud = g.get(regime + '_use_depth')
    ur = g.get(regime + '_use_rgb')
    losses, accs, grads = O.bank_nce_chunked(banks, g['idx'], xs, g['T'], use_depth=ud, use_rgb=ur, chunk=3)
    close(losses, g[regime + '_losses'], rtol=1e-5, atol=1e-6)
    close(accs, g[regime + '_accs'], rtol=0, atol=1e-4)
    for i in range(3):
        ref = g[regime + '_gx%d' % (i + 1)]
        assert rel_l2(grads[i], ref) < 1e-5 or float(ref.abs().max()) == 0 and float(grads[i].abs().max()) == 0


@pytest.mark.parametrize('regime', REGIMES)
def test_bank_nce(golden, regime):
    g = golden('bank_nce')
    banks = [g['bank0_%d' % i] for i in (1, 2, 3)]
    xs = [g['x%d' % i] for i in (1, 2, 3)]
    ud = g.get(regime + '_use_depth')
    ur = g.get(regime + '_use_rgb')
    losses, accs, grads, logits = O.bank_nce(banks, g['idx'], xs, g['T'], use_depth=ud, use_rgb=ur)
    assert torch.equal(g['idx'][:, 0], g['y'])
    for p in range(6):
        close(logits[p], g['logits%d' % p], rtol=1e-5, atol=2e-5)
    close(losses, g[regime + '_losses'], rtol=1e-5, atol=1e-6)
    close(accs, g[regime + '_accs'], rtol=0, atol=1e-4)
    for i in range(3):
        assert rel_l2(grads[i], g[regime + '_gx%d' % (i + 1)]) < 1e-5 or \
            float(g[regime + '_gx%d' % (i + 1)].abs().max()) == 0 and float(grads[i].abs().max()) == 0


def test_bank_update_duplicates_last_wins(golden):
    g = golden('bank_nce')
    winners = O.bank_update_winners(g['all_y'])
    assert winners.tolist() == [False, False, True, True, True, True, True, True, True, True, True, True]
    for i in (1, 2, 3):
        new = O.bank_update(g['bank0_%d' % i], g['all_x%d' % i], g['all_y'], g['m'])
        close(new, g['bank1_%d' % i], rtol=1e-6, atol=1e-7)
        touched = torch.zeros(g['n'], dtype=torch.bool)
        touched[g['all_y']] = True
        assert torch.equal(new[~touched], g['bank0_%d' % i][~touched])      # untouched rows bit-exact


def test_moco_queue(golden):
    g = golden('moco_queue')
    q1, q2 = g['queue0_1'], g['queue0_2']
    index = 0
    for s in range(3):
        l1 = O.moco_logits(g['s%d_q1' % s], g['s%d_k2' % s], q2, g['T'])
        l2 = O.moco_logits(g['s%d_q2' % s], g['s%d_k1' % s], q1, g['T'])
        close(l1, g['s%d_logits1' % s], atol=2e-5)
        close(l2, g['s%d_logits2' % s], atol=2e-5)
        q1, i1 = O.moco_enqueue(q1, g['s%d_all_k1' % s], index)
        q2, index = O.moco_enqueue(q2, g['s%d_all_k2' % s], index)
        assert i1 == index == g['s%d_index' % s]
        assert torch.equal(q1, g['s%d_queue_1' % s]) and torch.equal(q2, g['s%d_queue_2' % s])


def test_dense_soft_nce(golden):
    g = golden('dense_soft_nce')
    keep, m = O.dense_keep(g['depth_mask'], g['h'], g['h'])
    assert keep.tolist() == [True, True, True, False, True]
    ind = g['sample_ind']
    assert ind.shape == (4, g['S'])
    # every sampled pixel lies inside the resized mask of its image
    assert bool((torch.gather(m[keep], 1, ind) > 0).all())
    losses, accs, g1, g2 = O.dense_soft_nce(g['map1'], g['map2'], ind, keep, g['temperature'], g['use_depth'])
    close(losses, g['losses'], rtol=1e-5)
    close(accs, g['accs'], rtol=0, atol=1e-6)
    assert rel_l2(g1, g['grad_map1']) < 1e-5 and rel_l2(g2, g['grad_map2']) < 1e-5
    assert float(g1[3].abs().max()) == 0                                   # dropped image gets no gradient
    z, _, _, _ = O.dense_soft_nce(g['map1'], g['map2'], ind, keep, g['temperature'], torch.zeros(5))
    close(z, g['zero_losses'])


@pytest.mark.parametrize('J', [13, 16, 17])
def test_joint_nce(golden, J):
    g = golden('joint_nce')
    p = 'J%d_' % J
    losses, accs, g1, g2, g3 = O.joint_nce(g[p + 'map1'], g[p + 'map2'], g[p + 'feat3'], g[p + 'joints2d'],
                                           g[p + 'joints_vis'], g['temperature'], g[p + 'use_depth'])
    close(losses, g[p + 'losses'], rtol=1e-5)
    close(losses, g[p + 'losses_f64joints'], rtol=1e-5)
    close(accs, g[p + 'accs'], rtol=0, atol=1e-6)
    assert rel_l2(g1, g[p + 'grad_map1']) < 1e-5
    assert rel_l2(g2, g[p + 'grad_map2']) < 1e-5
    assert rel_l2(g3, g[p + 'grad_feat3']) < 1e-5
    pix = O.joint_pixels(g[p + 'joints2d'], 8)
    assert pix[0, 0] == 0 * 8 + 1 and pix[1, 2] == 7 * 8 + 0 and pix[2, 1] == pix[2, 0]


def test_joint_nce_all_ignored_is_nan(golden):
    g = golden('joint_nce')
    assert g['allignored_depth_loss_isnan']
    m = torch.randn(2, 128, 8, 8)
    losses, *_ = O.joint_nce(m, m, torch.randn(2, 16, 128), torch.rand(2, 16, 2) * 32,
                             torch.ones(2, 16).int(), 0.07, use_depth=torch.zeros(2))
    assert not math.isnan(float(losses[0])) and math.isnan(float(losses[1]))


def test_scl(golden):
    g = golden('scl')
    la, g1, g2, early = O.scl(g['map1'], g['map2'], g['joints2d'], g['temperature'], g['use_depth'], g['use_rgb'])
    assert not early
    close(la, torch.tensor(g['loss_with_rgb']), rtol=1e-5)
    assert rel_l2(g1, g['grad1_with_rgb']) < 1e-5 and rel_l2(g2, g['grad2_with_rgb']) < 1e-5
    lb, g1, g2, _ = O.scl(g['map1'], g['map2'], g['joints2d'], g['temperature'], g['use_depth'], None)
    close(lb, torch.tensor(g['loss_rgb_none']), rtol=1e-5)
    assert rel_l2(g1, g['grad1_rgb_none']) < 1e-5 and rel_l2(g2, g['grad2_rgb_none']) < 1e-5
    lc, _, _, early = O.scl(g['map1'], g['map2'], g['joints2d'], g['temperature'], torch.zeros(4), g['use_rgb'])
    assert early and float(lc) == 0 and g['n_early_out'] == 4 and float(g['early_out'].abs().max()) == 0
