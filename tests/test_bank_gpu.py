"""GPU parity of the memory-bank kernels through the C ABI:
HIP vs the CPU oracle on seeded inputs, vs the committed golden vectors generated from the
reference, and size-independent properties at BASELINE sizes."""
import pytest
import torch

from oracle import hcmoco_oracle as O

pytestmark = pytest.mark.gpu

# fp32 tolerances (SURVEY 8d "parity gate"): losses 1e-5 rel, grads 1e-4 rel-L2, logits 1e-5 abs
LOSS_RTOL, GRAD_REL_L2, LOGIT_ATOL = 1e-5, 1e-4, 1e-5


def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def ops():
    from hcmoco_amd import hip_ops
    return hip_ops


REGIMES = ['none', 'depth_mix', 'depth_all0', 'both_mix', 'both_none']


@pytest.mark.parametrize('regime', REGIMES)
def test_fused_vs_golden(golden, regime):
    g = golden('bank_nce')
    d = dev()
    banks = [g['bank0_%d' % i].to(d) for i in (1, 2, 3)]
    xs = [g['x%d' % i].to(d) for i in (1, 2, 3)]
    ud = g.get(regime + '_use_depth')
    ur = g.get(regime + '_use_rgb')
    losses, accs, gx = ops().bank_nce_fused_raw(banks, g['idx'].to(d), xs, g['T'],
                                                None if ud is None else ud.to(d),
                                                None if ur is None else ur.to(d))
    torch.cuda.synchronize()
    assert torch.allclose(losses.cpu(), g[regime + '_losses'], rtol=LOSS_RTOL, atol=1e-6), (losses, g[regime + '_losses'])
    assert torch.allclose(accs.cpu(), g[regime + '_accs'], rtol=0, atol=1e-4)
    for i in range(3):
        ref = g[regime + '_gx%d' % (i + 1)]
        if float(ref.abs().max()) == 0:
            assert float(gx[i].abs().max()) == 0
        else:
            assert rel_l2(gx[i], ref) < GRAD_REL_L2


def test_logits_api_vs_golden_and_backward(golden):
    g = golden('bank_nce')
    d = dev()
    banks = [g['bank0_%d' % i].to(d) for i in (1, 2, 3)]
    xs = [g['x%d' % i].to(d).requires_grad_(True) for i in (1, 2, 3)]
    logits = ops().bank_logits(xs, banks, g['idx'].to(d), g['T'])
    for p in range(6):
        assert torch.allclose(logits[p].cpu(), g['logits%d' % p], rtol=0, atol=LOGIT_ATOL)
    # CE through the materialised logits, autograd through the HIP backward
    tgt = torch.zeros(logits.shape[1], dtype=torch.long, device=d)
    loss = sum(torch.nn.functional.cross_entropy(logits[p], tgt) for p in range(6))
    loss.backward()
    for i in range(3):
        assert rel_l2(xs[i].grad, g['none_gx%d' % (i + 1)]) < GRAD_REL_L2


def test_autograd_total_matches_fused(golden):
    g = golden('bank_nce')
    d = dev()
    banks = [g['bank0_%d' % i].to(d) for i in (1, 2, 3)]
    xs = [g['x%d' % i].to(d).requires_grad_(True) for i in (1, 2, 3)]
    total, losses, accs = ops().bank_nce_fused(xs, banks, g['idx'].to(d), g['T'], g['depth_mix_use_depth'].to(d))
    (2.0 * total).backward()
    assert abs(float(total) - float(g['depth_mix_losses'].sum())) < 1e-4
    for i in range(3):
        assert rel_l2(xs[i].grad, 2.0 * g['depth_mix_gx%d' % (i + 1)]) < GRAD_REL_L2


def test_update_vs_golden_duplicates_last_wins(golden):
    g = golden('bank_nce')
    d = dev()
    banks = [g['bank0_%d' % i].to(d).clone() for i in (1, 2, 3)]
    ops().bank_update(banks, [g['all_x%d' % i].to(d) for i in (1, 2, 3)], g['all_y'].to(d), g['m'])
    torch.cuda.synchronize()
    touched = torch.zeros(g['n'], dtype=torch.bool)
    touched[g['all_y']] = True
    for i in range(3):
        new = banks[i].cpu()
        assert torch.equal(new[~touched], g['bank0_%d' % (i + 1)][~touched])     # bit-exact: untouched rows
        assert torch.allclose(new, g['bank1_%d' % (i + 1)], rtol=1e-6, atol=1e-7)   # the right writer won


def test_alias_tables_and_draw_bit_exact(golden):
    g = golden('alias_tables')
    prob, alias = ops().alias_build(g['probs'])
    assert torch.equal(prob, g['prob']) and torch.equal(alias, g['alias'])
    d = dev()
    B, K1 = 7, 333
    y = torch.arange(B, dtype=torch.int64) * 3
    idx = ops().alias_draw(prob.to(d), alias.to(d), y.to(d), B, K1, seed=0x1234567890abcdef, offset=987654321012)
    ref = O.alias_draw_philox(prob, alias, B * K1, 0x1234567890abcdef, 987654321012).view(B, K1)
    ref[:, 0] = y
    assert torch.equal(idx.cpu(), ref)
    # uniform table: every draw is kk itself (prob == 1) and inside [0, n)
    prob, alias = ops().alias_build(torch.ones(5000))
    idx = ops().alias_draw(prob.to(d), alias.to(d), None, 4, 10001, seed=5, offset=0)
    ref = O.alias_draw_philox(prob, alias, 4 * 10001, 5, 0).view(4, 10001)
    assert torch.equal(idx.cpu(), ref)
    assert int(idx.min()) >= 0 and int(idx.max()) < 5000


@pytest.mark.parametrize('B,K,n,D', [(3, 1, 50, 128), (5, 15, 64, 128), (4, 16, 64, 64), (2, 300, 512, 128),
                                     (33, 1000, 4096, 128), (8, 4097, 9000, 64)])
def test_fused_vs_oracle_ragged_sizes(B, K, n, D):
    """Edge sizes: K+1 not a multiple of the 16 streams, single chunk / many chunks, D=64."""
    torch.manual_seed(B * 1000 + K)
    d = dev()
    nrm = torch.nn.functional.normalize
    banks = [nrm(torch.randn(n, D)) for _ in range(3)]
    xs = [nrm(torch.randn(B, D)) for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1))
    ud = (torch.rand(B) < 0.7).long()
    ud[0] = 1
    lo, ao, go, logits_o = O.bank_nce(banks, idx, xs, 0.07, use_depth=ud)
    l, a, gx = ops().bank_nce_fused_raw([b.to(d) for b in banks], idx.to(d), [x.to(d) for x in xs], 0.07, ud.to(d))
    assert torch.allclose(l.cpu(), lo, rtol=LOSS_RTOL, atol=1e-6)
    assert torch.allclose(a.cpu(), ao, atol=1e-3)
    for i in range(3):
        assert rel_l2(gx[i], go[i]) < GRAD_REL_L2
    lg = ops().bank_logits([x.to(d) for x in xs], [b.to(d) for b in banks], idx.to(d), 0.07)
    for p in range(6):
        assert torch.allclose(lg[p].cpu(), logits_o[p], rtol=0, atol=LOGIT_ATOL)


def test_fused_unnormalised_inputs_are_stable():
    """Online softmax must not overflow when |logit| is large (rows / queries not unit norm)."""
    torch.manual_seed(3)
    d = dev()
    B, K, n, D = 4, 500, 1000, 128
    banks = [torch.randn(n, D) * 3 for _ in range(3)]
    xs = [torch.randn(B, D) * 3 for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1))
    lo, ao, go, _ = O.bank_nce([b.double() for b in banks], idx, [x.double() for x in xs], 0.07)
    l, a, gx = ops().bank_nce_fused_raw([b.to(d) for b in banks], idx.to(d), [x.to(d) for x in xs], 0.07)
    assert torch.isfinite(l).all()
    assert torch.allclose(l.cpu().double(), lo, rtol=LOSS_RTOL), (l, lo)
    for i in range(3):
        assert rel_l2(gx[i], go[i]) < GRAD_REL_L2, rel_l2(gx[i], go[i])


def test_full_size_properties():
    """BASELINE config sizes (B=32, K=16384, n=131072): size-independent checks.
    (1) fused loss == CE over the API-mode logits; (2) gradient rows are orthogonal-free sums:
    d/dx of sum(losses) contracted with x equals sum_p (E_p[l] - l_0)/|R| * T-free identity;
    (3) permuting the negatives leaves losses unchanged (softmax is permutation invariant)."""
    torch.manual_seed(0)
    d = dev()
    B, K, n, D, T = 32, 16384, 131072, 128, 0.07
    nrm = torch.nn.functional.normalize
    banks = [nrm(torch.randn(n, D, device=d)) for _ in range(3)]
    xs = [nrm(torch.randn(B, D, device=d)) for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1), device=d)
    l, a, gx = ops().bank_nce_fused_raw(banks, idx, xs, T)
    lg = ops().bank_logits(xs, banks, idx, T)
    tgt = torch.zeros(B, dtype=torch.long, device=d)
    for p in range(6):
        ce = torch.nn.functional.cross_entropy(lg[p].double(), tgt)
        assert abs(float(l[p]) - float(ce)) < 1e-5 * max(1.0, abs(float(ce)))
        acc = 100.0 * float((lg[p][:, 0] >= lg[p].max(1).values).float().mean())
        assert abs(float(a[p]) - acc) < 1e-3
    # <gx_a, x_a> = sum over the two pairs of a: mean_b( E_p[l] - l_0 )   (l is linear in x)
    for a_i, pairs in enumerate(((0, 4), (1, 2), (3, 5))):
        want = 0.0
        for p in pairs:
            pr = torch.softmax(lg[p].double(), 1)
            want += float(((pr * lg[p].double()).sum(1) - lg[p][:, 0].double()).mean())
        got = float((gx[a_i].double() * xs[a_i].double()).sum())
        assert abs(got - want) < 1e-4 * max(1.0, abs(want))
    perm = torch.cat([torch.zeros(1, dtype=torch.long, device=d), 1 + torch.randperm(K, device=d)])
    l2, _, gx2 = ops().bank_nce_fused_raw(banks, idx[:, perm].contiguous(), xs, T)
    assert torch.allclose(l, l2, rtol=1e-5)
    for i in range(3):
        assert rel_l2(gx2[i], gx[i]) < 1e-4


def test_moco_queue_vs_golden(golden):
    g = golden('moco_queue')
    d = dev()
    q1, q2 = g['queue0_1'].to(d).clone(), g['queue0_2'].to(d).clone()
    index = 0
    for s in range(3):
        l1 = ops().moco_logits(g['s%d_q1' % s].to(d), g['s%d_k2' % s].to(d), q2, g['T'])
        l2 = ops().moco_logits(g['s%d_q2' % s].to(d), g['s%d_k1' % s].to(d), q1, g['T'])
        assert torch.allclose(l1.cpu(), g['s%d_logits1' % s], atol=LOGIT_ATOL)
        assert torch.allclose(l2.cpu(), g['s%d_logits2' % s], atol=LOGIT_ATOL)
        i1 = ops().moco_enqueue(q1, g['s%d_all_k1' % s].to(d), index)
        index = ops().moco_enqueue(q2, g['s%d_all_k2' % s].to(d), index)
        assert i1 == index == g['s%d_index' % s]                         # bit-exact pointer
        assert torch.equal(q1.cpu(), g['s%d_queue_1' % s]) and torch.equal(q2.cpu(), g['s%d_queue_2' % s])


def test_cpu_tensors_fail_loudly():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops().bank_nce_fused_raw([torch.randn(8, 128)] * 3, torch.zeros(2, 3, dtype=torch.long),
                                 [torch.randn(2, 128)] * 3, 0.07)


# ----------------------------------------------------------------------------------------------
# bf16 bank storage (BASELINE config 5).  The oracle runs on the SAME bf16-rounded rows in fp32, so
# the tolerances stay the fp32 ones; only the update's final rounding is a bf16 ulp.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,K,n', [(4, 100, 300), (32, 4096, 20000)])
def test_bf16_banks_fused_logits_and_update(B, K, n):
    torch.manual_seed(B + K)
    d = dev()
    D, T, mom = 128, 0.07, 0.5
    nrm = torch.nn.functional.normalize
    banks16 = [nrm(torch.randn(n, D)).to(torch.bfloat16) for _ in range(3)]
    banks32 = [b.float() for b in banks16]
    xs = [nrm(torch.randn(B, D)) for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1))
    ud = (torch.rand(B) < 0.7).long()
    ud[0] = 1
    lo, ao, go, logits_o = O.bank_nce(banks32, idx, xs, T, use_depth=ud)
    gb = [b.to(d) for b in banks16]
    l, a, gx = ops().bank_nce_fused_raw(gb, idx.to(d), [x.to(d) for x in xs], T, ud.to(d))
    assert torch.allclose(l.cpu(), lo, rtol=LOSS_RTOL, atol=1e-6)
    assert torch.allclose(a.cpu(), ao, atol=1e-3)
    for i in range(3):
        assert rel_l2(gx[i], go[i]) < GRAD_REL_L2
    xg = [x.to(d).requires_grad_(True) for x in xs]
    lg = ops().bank_logits(xg, gb, idx.to(d), T)
    for p in range(6):
        assert torch.allclose(lg[p].cpu(), logits_o[p], rtol=0, atol=LOGIT_ATOL)
    lg.sum().backward()
    xo = [x.clone().requires_grad_(True) for x in xs]
    sum(t.sum() for t in O.bank_logits(banks32, idx, xo, T)).backward()
    for i in range(3):
        assert rel_l2(xg[i].grad, xo[i].grad) < GRAD_REL_L2
    # update: fp32 math on the bf16 rows, result rounded to nearest-even bf16; last duplicate wins
    all_y = torch.randint(0, n, (2 * B,))
    all_y[-1] = all_y[0]
    all_x = [nrm(torch.randn(2 * B, D)) for _ in range(3)]
    ops().bank_update(gb, [x.to(d) for x in all_x], all_y.to(d), mom)
    torch.cuda.synchronize()
    touched = torch.zeros(n, dtype=torch.bool)
    touched[all_y] = True
    for i in range(3):
        new = gb[i].cpu()
        assert new.dtype == torch.bfloat16
        assert torch.equal(new[~touched].view(torch.int16), banks16[i][~touched].view(torch.int16))   # bit-exact
        ref = O.bank_update(banks32[i], all_x[i], all_y, mom)
        err = (new.float() - ref).abs()
        assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all())                                       # <= 1 bf16 ulp
        assert float((new.float()[touched].norm(dim=1) - 1).abs().max()) < 1e-2


def test_config3_size_k65536_properties():
    """BASELINE config 3 size (K=65536): fused loss == CE over API-mode logits, and doubling the
    negatives by repeating them raises every loss by exactly the log-sum-exp identity
    lse(l ++ l_neg) = log(exp(lse(l)) + sum exp(l_neg))."""
    torch.manual_seed(1)
    d = dev()
    B, K, n, D, T = 32, 65536, 131072, 128, 0.07
    nrm = torch.nn.functional.normalize
    banks = [nrm(torch.randn(n, D, device=d)) for _ in range(3)]
    xs = [nrm(torch.randn(B, D, device=d)) for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1), device=d)
    l, a, gx = ops().bank_nce_fused_raw(banks, idx, xs, T)
    lg = ops().bank_logits(xs, banks, idx, T)
    tgt = torch.zeros(B, dtype=torch.long, device=d)
    for p in range(6):
        assert abs(float(l[p]) - float(torch.nn.functional.cross_entropy(lg[p].double(), tgt))) < LOSS_RTOL * float(l[p])
    idx2 = torch.cat([idx, idx[:, 1:]], dim=1).contiguous()               # negatives twice
    l2, _, _ = ops().bank_nce_fused_raw(banks, idx2, xs, T)
    for p in range(6):
        lse1 = torch.logsumexp(lg[p].double(), 1)
        lse_neg = torch.logsumexp(lg[p][:, 1:].double(), 1)
        want = (torch.logaddexp(lse1, lse_neg) - lg[p][:, 0].double()).mean()
        assert abs(float(l2[p]) - float(want)) < LOSS_RTOL * float(want)


# ----------------------------------------------------------------------------------------------
# BASELINE sizes against the ORACLE (not against another HIP path): the oracle is evaluated one sample at a
# time (each call gathers (K+1) x 128 x 3 rows, trivial on the host) and the per-sample terms are composed
# into the masked means of _compute_loss_accuracy (contrast_trainer.py:212-253) exactly as O.bank_nce does.
# ----------------------------------------------------------------------------------------------
def _oracle_per_sample(banks, idx, xs, T, use_depth=None, use_rgb=None):
    B = idx.shape[0]
    per_l = torch.zeros(B, 6, dtype=torch.float64)
    per_ok = torch.zeros(B, 6, dtype=torch.float64)
    per_g = [torch.zeros(B, 6, xs[0].shape[1], dtype=torch.float64) for _ in range(3)]
    for b in range(B):
        logits = O.bank_logits(banks, idx[b:b + 1], [x[b:b + 1] for x in xs], T)
        rows = [bk.index_select(0, idx[b]).double() for bk in banks]
        for p, (a, c) in enumerate(O.PAIRS):
            l = logits[p][0].double()
            per_l[b, p] = torch.logsumexp(l, 0) - l[0]
            per_ok[b, p] = float(l[0] >= l.max())
            per_g[a][b, p] = (torch.softmax(l, 0) @ rows[c] - rows[c][0]) / T
    sel, deg = O.bank_row_sets(B, use_depth, use_rgb)
    losses, accs = torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
    grads = [torch.zeros(B, xs[0].shape[1], dtype=torch.float64) for _ in range(3)]
    for p, (a, c) in enumerate(O.PAIRS):
        if deg[p]:
            continue
        R = sel[p]
        cnt = int(R.sum())
        losses[p] = per_l[R, p].sum() / cnt
        accs[p] = 100.0 * per_ok[R, p].sum() / cnt
        grads[a] += per_g[a][:, p] * R.double().unsqueeze(1) / cnt
    return losses, accs, grads


@pytest.mark.parametrize('K,dtype,masked', [(16384, torch.float32, True), (16384, torch.float32, False),
                                            (65536, torch.float32, True), (131072, torch.bfloat16, True)])
def test_baseline_sizes_against_the_oracle(K, dtype, masked):
    """B=32, n=131072, D=128 at K=16384 (config 2), K=65536 (config 3) and bf16 banks at K=131072 (config 5):
    all six losses, accuracies and all 32 gradient rows per modality against the oracle, plus the momentum
    update of the touched rows.  bf16: the oracle reads the same bf16-rounded rows in fp32."""
    torch.manual_seed(K)
    d = dev()
    B, n, D, T, mom = 32, 131072, 128, 0.07, 0.5
    nrm = torch.nn.functional.normalize
    banks_s = [nrm(torch.randn(n, D)).to(dtype) for _ in range(3)]
    banks_o = [b.float() for b in banks_s]
    xs = [nrm(torch.randn(B, D)) for _ in range(3)]
    y = torch.randperm(n)[:B]
    idx = torch.randint(0, n, (B, K + 1))
    idx[:, 0] = y
    # make a few positives genuinely the arg-max so that the accuracies are not all zero
    for b in range(0, B, 5):
        for m in range(3):
            banks_o[m][y[b]] = nrm(xs[(m + 1) % 3][b] + 0.05 * torch.randn(D), dim=0)
            banks_s[m][y[b]] = banks_o[m][y[b]].to(dtype)
            banks_o[m][y[b]] = banks_s[m][y[b]].float()
    ud = None
    if masked:
        ud = (torch.rand(B) < 0.75).long()
        ud[0] = 1
    lo, ao, go = _oracle_per_sample(banks_o, idx, xs, T, use_depth=ud)
    gb = [b.to(d) for b in banks_s]
    l, a, gx = ops().bank_nce_fused_raw(gb, idx.to(d), [x.to(d) for x in xs], T, None if ud is None else ud.to(d))
    torch.cuda.synchronize()
    assert torch.allclose(l.cpu().double(), lo, rtol=LOSS_RTOL, atol=1e-6), (l, lo)
    assert torch.allclose(a.cpu().double(), ao, atol=1e-3), (a, ao)
    assert float(ao.max()) > 0
    for i in range(3):
        assert rel_l2(gx[i], go[i]) < GRAD_REL_L2, (i, rel_l2(gx[i], go[i]))
    # momentum update after the reads (rank-major, last duplicate wins), oracle on the touched rows only
    all_y = torch.cat([y, y[:3], torch.randint(0, n, (B - 3,))])
    all_x = [nrm(torch.randn(2 * B, D)) for _ in range(3)]
    before = [b.clone() for b in gb]
    ops().bank_update(gb, [x.to(d) for x in all_x], all_y.to(d), mom)
    torch.cuda.synchronize()
    touched = torch.zeros(n, dtype=torch.bool)
    touched[all_y] = True
    rows = touched.nonzero().flatten()
    for i in range(3):
        new = gb[i].cpu()
        assert torch.equal(new[~touched].view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                           before[i].cpu()[~touched].view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
        ref = O.bank_update(banks_o[i], all_x[i], all_y, mom)[rows]
        err = (new[rows].float() - ref).abs()
        if dtype == torch.bfloat16:
            assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-6).all())
        else:
            assert float(err.max()) <= 1e-6


# ----------------------------------------------------------------------------------------------
# The lean pass (csrc/bank_lean.hip) starts its exponentials at the bound 1.01 |x| log2(e) / T, which unit bank rows
# cannot exceed.  Inputs that break that assumption take its other paths: rows longer than 1 (a logit above the reference
# point: the merged online-softmax rescale), queries so long that the bound is useless (reference point "minus infinity":
# the branch chases the maximum), and both at once.  Same oracle, same fp32 tolerances.
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('case', ['long_rows', 'long_queries', 'both', 'mixed_batch', 'tiny_T'])
def test_lean_pass_outside_the_unit_row_assumption(case, dtype):
    torch.manual_seed(len(case) * 7 + len(dtype))
    d = dev()
    B, K, n, D, T = 6, 1500, 5000, 128, 0.07
    nrm = torch.nn.functional.normalize
    banks = [nrm(torch.randn(n, D)) for _ in range(3)]
    xs = [nrm(torch.randn(B, D)) for _ in range(3)]
    if case in ('long_rows', 'both'):
        for b in banks:
            b[::7] *= 1.8                      # a seventh of the rows exceeds the bound by far
            b[3::11] *= 1.02                   # and some only just
    if case in ('long_queries', 'both'):
        xs = [x * 3.5 for x in xs]             # bound above 60 log2 units: chase mode
    if case == 'mixed_batch':
        for x in xs:
            x[1] *= 4.0                        # one sample in chase mode, its neighbours at the bound
            x[4] *= 0.05
        banks[1][5::13] *= 1.5
    if case == 'tiny_T':
        T = 0.02                               # |x| log2(e) / T = 72: chase mode for unit queries
    if dtype == 'bf16':
        banks = [b.to(torch.bfloat16) for b in banks]
    banks32 = [b.float() for b in banks]
    idx = torch.randint(0, n, (B, K + 1))
    ud = torch.tensor([1, 0, 1, 1, 0, 1])
    lo, ao, go, _ = O.bank_nce(banks32, idx, xs, T, use_depth=ud)
    l, a, gx = ops().bank_nce_fused_raw([b.to(d) for b in banks], idx.to(d), [x.to(d) for x in xs], T, ud.to(d))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(l).all())
    assert torch.allclose(l.cpu(), lo, rtol=LOSS_RTOL, atol=1e-6), (l.cpu(), lo)
    assert torch.allclose(a.cpu(), ao, atol=1e-3)
    for i in range(3):
        assert rel_l2(gx[i], go[i]) < GRAD_REL_L2, (i, rel_l2(gx[i], go[i]))
