"""Error study for VERDICT r04 #4 (CPU only): the dense soft-InfoNCE and the SCL similarity / gradient contractions on
the bf16 matrix cores with fp32-level accuracy through operand splitting.

    x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (8 + 8 + 8 mantissa bits)
    3-term:  x.y ~ hi.hi + hi.mid + mid.hi                    drops terms ~2^-16 |x||y|
    4-term:  + mid.mid                                        (the one dropped term that is systematic on a diagonal x.x)
    6-term:  + mid.mid + hi.lo + lo.hi                        drops terms ~2^-24 |x||y|

Every partial product of two bf16 values is exact in fp32 (16-bit mantissa product), the matrix core accumulates in
fp32: emulated here as fp32 matmuls of the bf16-valued parts (torch CPU fp32 GEMM, fp32 accumulate), summed in fp32.
Gate: losses 1e-5 relative, gradients 1e-4 relative L2, against the float64 evaluation of the same math;
the plain fp32 contraction (what v_mfma_f32_16x16x4_f32 computes) and the one-term bf16 contraction (config 5) are listed
beside it.  Cases: golden/dense_soft_nce.npz, golden/scl.npz (the reference's own inputs) and BASELINE-size synthetic
maps (B = 32, S = 400 / J = 17, C = 128, tau = 0.07).  usage: python tools/studies/split_bf16_error.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def split(x, levels):
    parts, r = [], x.clone()
    for _ in range(levels):
        p = r.to(torch.bfloat16).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


TERMS = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 4: [(0, 0), (0, 1), (1, 0), (1, 1)], 6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}


def contract(a, b, mode):
    """a [.., M, K] @ b [.., K, N] under ``mode``: 'f64', 'f32', 1, 3 or 6 (bf16 terms)."""
    if mode == 'f64':
        return a.double() @ b.double()
    if mode == 'f32':
        return (a.float() @ b.float()).double()
    pa, pb = split(a.float(), 3), split(b.float(), 3)
    # smallest terms first, like an accumulator that receives hi.hi last would -- order is not material at this size
    acc = torch.zeros(a.shape[:-1] + b.shape[-1:], dtype=torch.float32)
    for i, j in reversed(TERMS[mode]):
        acc = acc + pa[i] @ pb[j]
    return acc.double()


def dense_loss(f1, f2, tgt, tau, mode):
    """f1, f2 [B, S, C] unit rows (fp32), tgt [B, S, S] column-normalised soft target.  -> (losses[2], df1, df2)"""
    A = contract(f2, f1.transpose(1, 2), mode) / tau
    ls_col, ls_row = torch.log_softmax(A, 1), torch.log_softmax(A, 2)
    t = tgt.double()
    l1 = -(t * ls_col).sum(1).mean()
    l2 = -(t.transpose(1, 2) * ls_row).sum(2).mean()
    B, S = f1.shape[:2]
    G = ((ls_col.exp() - t) + (ls_row.exp() - t.transpose(1, 2))) / (B * S)
    Gm = G if mode == 'f64' else G.float()
    df2 = contract(Gm, f1, mode) / tau
    df1 = contract(Gm.transpose(1, 2), f2, mode) / tau
    return torch.stack([l1, l2]), df1, df2


def scl_loss(x, jid, valid, tau, mode):
    """x [N, C] unit rows; positives: same joint id, other row, both valid.  -> (loss, dx)"""
    N = x.shape[0]
    L = contract(x, x.t(), mode) / tau
    lp = torch.log_softmax(L, 1)
    pos = (jid[:, None] == jid[None, :]) & ~torch.eye(N, dtype=torch.bool) & valid[:, None] & valid[None, :]
    cnt = pos.sum(1).clamp_min(1).double()
    loss = (-(lp * pos).sum(1) / cnt).mean()
    # d loss / d L
    w = pos.double() / cnt[:, None]
    G = (lp.exp() * w.sum(1, keepdim=True) - w) / N
    Gs = G + G.t()
    Gm = Gs if mode == 'f64' else Gs.float()
    return loss, contract(Gm, x, mode) / tau


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def report(name, fn):
    ref = fn('f64')
    print('%-44s %24s %12s %12s' % (name, 'mode', 'loss rel', 'grad rel-L2'))
    for mode in ('f32', 1, 3, 4, 6):
        got = fn(mode)
        lr = float(((got[0] - ref[0]).abs() / ref[0].abs().clamp_min(1e-30)).max())
        gr = max(rel(g, r) for g, r in zip(got[1:], ref[1:]))
        tag = {'f32': 'fp32 (today)', 1: 'bf16 1-term', 3: 'split 3-term', 4: 'split 4-term (+ mid.mid)', 6: 'split 6-term'}[mode]
        ok = 'PASS' if lr <= 1e-5 and gr <= 1e-4 else 'fail'
        print('%-44s %24s %12.3e %12.3e   %s (gate 1e-5 / 1e-4)' % ('', tag, lr, gr, ok))


def soft_target(ind, w):
    q = torch.stack([ind // w, ind % w], -1).float()
    dist = torch.sqrt(((q[:, :, None, :] - q[:, None, :, :]) ** 2).sum(-1))
    return torch.softmax(-dist, 1)


def main():
    torch.manual_seed(0)
    tau = 0.07
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'dense_soft_nce.npz'))
    from oracle import hcmoco_oracle as O
    m1, m2 = torch.from_numpy(g['map1']), torch.from_numpy(g['map2'])
    ind = torch.from_numpy(g['sample_ind'])
    keep, _ = O.dense_keep(torch.from_numpy(g['depth_mask']), int(g['h']), int(g['h']))
    kept = torch.nonzero(keep).flatten()
    f1 = F.normalize(O.gather_pixels(m1[kept], ind), dim=-1)
    f2 = F.normalize(O.gather_pixels(m2[kept], ind), dim=-1)
    tgt = soft_target(ind, m1.shape[-1])
    report('dense, golden (B=%d S=%d)' % tuple(f1.shape[:2]), lambda mode: dense_loss(f1, f2, tgt, tau, mode))

    for seed, corr in ((1, 0.0), (2, 0.9)):
        # BASELINE size; corr: how alike the two modalities' rows are (trained encoders make them alike, which sharpens
        # the softmax -- the regime where a logit error matters most)
        gen = torch.Generator().manual_seed(seed)
        B, S, C, w = 32, 400, 128, 64
        base = torch.randn(B, S, C, generator=gen)
        f1 = F.normalize(base + (1 - corr) * torch.randn(B, S, C, generator=gen), dim=-1)
        f2 = F.normalize(base + (1 - corr) * torch.randn(B, S, C, generator=gen), dim=-1)
        ind = torch.randint(0, w * w, (B, S), generator=gen)
        tgt = soft_target(ind, w)
        report('dense, BASELINE size, row correlation %.1f' % corr, lambda mode: dense_loss(f1, f2, tgt, tau, mode))

    for seed, corr, J in ((3, 0.0, 17), (4, 0.9, 17), (5, 0.9, 16)):
        gen = torch.Generator().manual_seed(seed)
        B, C = 32, 128
        N = 2 * B * J
        jid = torch.arange(J).repeat(2 * B)
        proto = torch.randn(J, C, generator=gen)
        x = F.normalize(corr * proto[jid] + (1 - corr * 0.5) * torch.randn(N, C, generator=gen), dim=-1)
        valid = torch.rand(N, generator=gen) < 0.8
        report('SCL, N = 2*32*%d, joint-prototype weight %.1f' % (J, corr), lambda mode: scl_loss(x, jid, valid, tau, mode))


if __name__ == '__main__':
    main()
