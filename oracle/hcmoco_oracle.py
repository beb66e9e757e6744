"""CPU oracle for the HCMoCo contrastive hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch on CPU tensors; closed-form backward, no
autograd) of the reference algorithm for the rows of SURVEY.md section 8a.  It
exists to *check* the HIP kernels; it is never the thing shipped or measured:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product package ``hcmoco_amd`` never does.

Parity is PINNED: every function below is checked against golden vectors that
``tests/golden/gen_golden.py`` produced by running the reference's own code
(``/root/reference/pycontrast``) in the build container
(``tests/test_oracle_vs_golden.py``).

Reference citations are relative to ``/root/reference/pycontrast``.
All floating point is fp32 unless ``dtype`` says otherwise.
"""
import numpy as np
import torch
import torch.nn.functional as F

PAIRS = ((0, 1), (1, 0), (1, 2), (2, 1), (0, 2), (2, 0))
"""(query modality a, bank modality c) for logits12,21,23,32,13,31
(memory/mem_bank.py:186-191)."""

_M32 = np.uint64(0xFFFFFFFF)


# --------------------------------------------------------------------------- #
# row 1 -- Walker alias tables + draw      memory/alias_multinomial.py:7-65
# --------------------------------------------------------------------------- #
def alias_build(probs):
    """Alias tables exactly as ``AliasMethod.__init__`` builds them
    (alias_multinomial.py:7-42): fp32 table, LIFO work lists, leftovers -> 1."""
    probs = torch.as_tensor(probs, dtype=torch.float32).clone()
    if probs.sum() > 1:
        probs = probs / probs.sum()
    K = probs.numel()
    prob = (probs * K).to(torch.float32).numpy().copy()      # fp32 K*prob  (:20)
    alias = np.zeros(K, dtype=np.int64)
    one = np.float32(1.0)
    smaller = [k for k in range(K) if prob[k] < one]
    larger = [k for k in range(K) if not (prob[k] < one)]
    while smaller and larger:
        small = smaller.pop()
        large = larger.pop()
        alias[small] = large
        prob[large] = np.float32(np.float32(prob[large] - one) + prob[small])   # (:35)
        if prob[large] < one:
            smaller.append(large)
        else:
            larger.append(large)
    for k in smaller + larger:
        prob[k] = one
    return torch.from_numpy(prob), torch.from_numpy(alias)


def _mulhilo32(a, b):
    p = a.astype(np.uint64) * np.uint64(b)
    return (p >> np.uint64(32)).astype(np.uint32), (p & _M32).astype(np.uint32)


def philox4x32_10(ctr, key):
    """Philox-4x32-10 (Salmon et al., SC'11).  ``ctr``: uint32 [N,4], ``key``: (k0,k1)."""
    c = [ctr[:, i].astype(np.uint32).copy() for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over='ignore'):
        for _ in range(10):
            hi0, lo0 = _mulhilo32(c[0], 0xD2511F53)
            hi1, lo1 = _mulhilo32(c[2], 0xCD9E8D57)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return np.stack(c, axis=1)


def alias_draw_philox(prob, alias, N, seed, offset):
    """Counter-based restatement of ``AliasMethod.draw`` (alias_multinomial.py:48-65).

    Element i uses Philox counter (i, 0, offset_lo, offset_hi), key = seed:
      kk = ((r0 << 32 | r1) mod n)                     -- ``random_(0, K)``
      b  = (r2 >> 8) * 2^-24 < prob[kk]                -- ``bernoulli(prob[kk])``
      out = kk if b else alias[kk]                     -- ``kk*b + alias*(1-b)``
    The reference consumes torch's global generator instead; the *distribution*
    and the arithmetic downstream of the uniforms are identical, and the HIP
    kernel ``hcm_alias_draw`` is bit-exact against this function."""
    prob = np.asarray(prob, dtype=np.float32)
    alias = np.asarray(alias, dtype=np.int64)
    n = prob.shape[0]
    i = np.arange(N, dtype=np.uint64)
    ctr = np.stack([(i & _M32), (i >> np.uint64(32)),
                    np.full(N, offset & 0xFFFFFFFF, np.uint64),
                    np.full(N, (offset >> 32) & 0xFFFFFFFF, np.uint64)], axis=1).astype(np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    r64 = (r[:, 0].astype(np.uint64) << np.uint64(32)) | r[:, 1].astype(np.uint64)
    kk = (r64 % np.uint64(n)).astype(np.int64)
    u = (r[:, 2] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    b = u < prob[kk]
    return torch.from_numpy(np.where(b, kk, alias[kk]))


# --------------------------------------------------------------------------- #
# rows 2+4 -- bank logits, CE, accuracy, backward
#             memory/mem_bank.py:172-193, :30-40 ; learning/contrast_trainer.py:212-253
# --------------------------------------------------------------------------- #
def bank_logits(banks, idx, xs, T):
    """Six logit sets [B,K+1] (mem_bank.py:179-191): ``bmm(w_c, x_a)/T``."""
    out = []
    gathered = [b.index_select(0, idx.reshape(-1)).view(idx.shape[0], idx.shape[1], -1) for b in banks]
    for a, c in PAIRS:
        out.append(torch.bmm(gathered[c], xs[a].unsqueeze(2)).squeeze(2) / T)
    return out


def bank_row_sets(B, use_depth=None, use_rgb=None):
    """Row selection of ``_compute_loss_accuracy`` (contrast_trainer.py:223-250).

    Returns (sel[6,B] bool, degenerate[6] bool).  ``degenerate[p]`` marks a set whose
    loss is ``(l-l).sum() == 0`` with accuracy 0 (the zero-sample early return
    :228-231 / :236-239)."""
    sel = torch.ones(6, B, dtype=torch.bool)
    deg = torch.zeros(6, dtype=torch.bool)
    if use_rgb is not None:
        assert use_depth is not None
        both = (torch.as_tensor(use_depth) == 1) & (torch.as_tensor(use_rgb) == 1)
        if both.sum() == 0:
            deg[:4] = True
            sel[:4] = False
        else:
            sel[:] = both
    elif use_depth is not None:
        ud = torch.as_tensor(use_depth)
        if ud.sum() == 0:
            deg[:4] = True
            sel[:4] = False
        else:
            sel[:4] = (ud == 1)
    return sel, deg


def bank_nce(banks, idx, xs, T, use_depth=None, use_rgb=None):
    """Fused statement of rows 2+4: losses[6], accs[6] (percent), grads wrt x1..3.

    loss_p = mean_{b in R_p}(logsumexp_k l - l_0); acc_p = 100*|{argmax==0}|/|R_p|
    (torch.topk picks the lowest index on ties, learning/util.py:24-38);
    d loss_p / d x_a[b] = (sum_k p_k M_c[r_k] - M_c[r_0]) / (T*|R_p|)   (SURVEY appendix A.1)."""
    B = idx.shape[0]
    logits = bank_logits(banks, idx, xs, T)
    sel, deg = bank_row_sets(B, use_depth, use_rgb)
    losses = torch.zeros(6, dtype=xs[0].dtype)
    accs = torch.zeros(6, dtype=xs[0].dtype)
    grads = [torch.zeros_like(x) for x in xs]
    gathered = [b.index_select(0, idx.reshape(-1)).view(B, idx.shape[1], -1) for b in banks]
    for p, (a, c) in enumerate(PAIRS):
        if deg[p]:
            continue
        R = sel[p]
        cnt = int(R.sum())
        l = logits[p]
        lse = torch.logsumexp(l, dim=1)
        per = lse - l[:, 0]
        losses[p] = per[R].sum() / cnt
        correct = l[:, 0] >= l.max(dim=1).values
        accs[p] = 100.0 * correct[R].sum().to(xs[0].dtype) / cnt
        prob = torch.softmax(l, dim=1)
        g = torch.bmm(prob.unsqueeze(1), gathered[c]).squeeze(1) - gathered[c][:, 0]
        g = g / (T * cnt)
        g[~R] = 0
        grads[a] = grads[a] + g
    return losses, accs, grads, logits


def bank_nce_chunked(banks, idx, xs, T, use_depth=None, use_rgb=None, chunk=4):
    """``bank_nce`` (same definitions, same masked means) evaluated ``chunk`` samples at a time in float64, so
    that BASELINE sizes (B=32, K up to 131072: 3 x 2.1 GB of gathered rows in one piece) stay small on the host.
    Returns (losses[6], accs[6], grads[3]) in float64."""
    B, D = xs[0].shape
    per_l = torch.zeros(B, 6, dtype=torch.float64)
    per_ok = torch.zeros(B, 6, dtype=torch.float64)
    per_g = [torch.zeros(B, 6, D, dtype=torch.float64) for _ in range(3)]
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        rows = [bk.index_select(0, idx[s:e].reshape(-1)).view(e - s, idx.shape[1], D).double() for bk in banks]
        for p, (a, c) in enumerate(PAIRS):
            l = torch.bmm(rows[c].to(xs[a].dtype), xs[a][s:e].unsqueeze(2)).squeeze(2).double() / T   # fp32 products like bank_logits
            per_l[s:e, p] = torch.logsumexp(l, 1) - l[:, 0]
            per_ok[s:e, p] = (l[:, 0] >= l.max(1).values).double()
            per_g[a][s:e, p] = (torch.bmm(torch.softmax(l, 1).unsqueeze(1), rows[c]).squeeze(1) - rows[c][:, 0]) / T
    sel, deg = bank_row_sets(B, use_depth, use_rgb)
    losses, accs = torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
    grads = [torch.zeros(B, D, dtype=torch.float64) for _ in range(3)]
    for p, (a, c) in enumerate(PAIRS):
        if deg[p]:
            continue
        R = sel[p]
        cnt = int(R.sum())
        losses[p] = per_l[R, p].sum() / cnt
        accs[p] = 100.0 * per_ok[R, p].sum() / cnt
        grads[a] += per_g[a][:, p] * R.double().unsqueeze(1) / cnt
    return losses, accs, grads


# --------------------------------------------------------------------------- #
# row 3 -- momentum update of the bank        memory/mem_bank.py:15-28
# --------------------------------------------------------------------------- #
def bank_update(bank, all_x, all_y, m):
    """``w = m*mem[y] + (1-m)*x ; w /= max(|w|,1e-12) ; mem[y] = w`` with every read
    taken from the PRE-update bank and duplicates resolved last-wins in gather
    (rank-major) order -- torch CPU ``index_copy_`` semantics."""
    old = bank.index_select(0, all_y)
    w = old * m + all_x * (1 - m)
    w = F.normalize(w)
    out = bank.clone()
    for j in range(all_y.numel()):          # sequential => last wins
        out[all_y[j]] = w[j]
    return out


def bank_update_winners(all_y):
    """Bit-exact bookkeeping: j is the writer of row y[j] iff no later j' has y[j']==y[j]."""
    y = all_y.tolist()
    last = {}
    for j, v in enumerate(y):
        last[v] = j
    return torch.tensor([last[v] == j for j, v in enumerate(y)], dtype=torch.bool)


# --------------------------------------------------------------------------- #
# MoCo queue (secondary)                      memory/mem_moco.py:6-49, 91-142
# --------------------------------------------------------------------------- #
def moco_logits(q, k, queue, T):
    pos = (q * k).sum(1, keepdim=True)
    neg = q @ queue.t()
    return torch.cat([pos, neg], dim=1) / T


def moco_enqueue(queue, all_k, index):
    """``queue[(index + arange(n)) % K] = all_k`` ; pointer advances by n mod K."""
    K = queue.shape[0]
    ids = (torch.arange(all_k.shape[0]) + index) % K
    out = queue.clone()
    for j in range(ids.numel()):
        out[ids[j]] = all_k[j]
    return out, (index + all_k.shape[0]) % K


# --------------------------------------------------------------------------- #
# helpers shared by rows 5-7
# --------------------------------------------------------------------------- #
def _normalize_bwd(g, x, eps=1e-12):
    """Backward of ``F.normalize(x, dim=-1)``: (g - (g.xhat) xhat)/max(|x|,eps)."""
    nrm = x.norm(dim=-1, keepdim=True).clamp_min(eps)
    xh = x / nrm
    return (g - (g * xh).sum(-1, keepdim=True) * xh) / nrm


def nearest_resize_mask(mask, h, w):
    """``F.interpolate(mask[:,None].float(), size=(h,w), mode='nearest')``
    (contrast_trainer.py:672): src = floor(dst * in/out)."""
    B, H, W = mask.shape
    ri = torch.floor(torch.arange(h, dtype=torch.float32) * (H / h)).long().clamp_max(H - 1)
    ci = torch.floor(torch.arange(w, dtype=torch.float32) * (W / w)).long().clamp_max(W - 1)
    return mask.float()[:, ri][:, :, ci]


def joint_pixels(joints2d, h):
    """``(r,c) = clamp(floor(joints2d / 4), 0, h-1)`` ; flat = r*h + c
    (contrast_trainer.py:757-761; ``//`` is floor division on floats)."""
    d = torch.floor(torch.as_tensor(joints2d) / 4).long()
    d = d.clamp(0, h - 1)
    return d[..., 0] * h + d[..., 1]


def gather_pixels(fmap, ind):
    """fmap [B,C,h,w] (any strides), ind [B,S] flat pixel -> [B,S,C]."""
    B, C = fmap.shape[:2]
    flat = fmap.reshape(B, C, -1)
    return torch.gather(flat, 2, ind.unsqueeze(1).expand(B, C, ind.shape[1])).permute(0, 2, 1)


def scatter_pixels(g, ind, shape):
    """Backward of ``gather_pixels``: duplicate pixels accumulate (torch.gather backward)."""
    B, C, h, w = shape
    out = torch.zeros(B, C, h * w, dtype=g.dtype)
    out.scatter_add_(2, ind.unsqueeze(1).expand(B, C, ind.shape[1]), g.permute(0, 2, 1))
    return out.view(B, C, h, w)


def pixel_sample_philox(depth_mask, h, w, S, use_depth, seed, offset):
    """Counter-based restatement of the pixel sampling of ``_compute_soft_pri3d_loss_accuracy``
    (contrast_trainer.py:671-685): nearest-resize the mask, keep images whose resized mask is non-empty (and, :663-665,
    only if some sample has depth), draw S pixels per kept image uniformly WITH replacement from the valid ones
    (``torch.multinomial(mask, S, replacement=True)`` on a 0/1 mask).  Element e = b*S + s uses Philox counter
    (e_lo, e_hi, offset_lo, offset_hi), key = seed; k = (r0 * count) >> 32; pixel = the (k+1)-th valid one in raster
    order.  The reference consumes torch's generator instead; the distribution is the same and ``hcm_pixel_sample``
    is bit-exact against this function.  -> (sample_ind [B,S] int64 (0 for dropped images), keep [B] bool)."""
    B = depth_mask.shape[0]
    m = nearest_resize_mask(depth_mask, h, w).reshape(B, h * w) > 0
    cnt = m.long().cumsum(1)
    total = cnt[:, -1]
    any_depth = True if use_depth is None else bool(torch.as_tensor(use_depth).sum() > 0)
    keep = (total > 0) & any_depth
    N = B * S
    i = np.arange(N, dtype=np.uint64)
    ctr = np.stack([(i & _M32), (i >> np.uint64(32)),
                    np.full(N, offset & 0xFFFFFFFF, np.uint64),
                    np.full(N, (offset >> 32) & 0xFFFFFFFF, np.uint64)], axis=1).astype(np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    r0 = torch.from_numpy(r[:, 0].astype(np.int64)).view(B, S)
    k = (r0 * total.view(B, 1)) >> 32
    ind = torch.searchsorted(cnt, k, right=True).clamp_max(h * w - 1)       # smallest q with cnt[q] > k
    ind = torch.where(keep.view(B, 1), ind, torch.zeros_like(ind))
    return ind, keep


def heads(maps1, maps2, feat3, W, b):
    """Rows 9: global average pool of the four branches + concat, Linear, L2-normalise for the two HRNets; mean over
    the joints, Linear, L2-normalise for the SemGCN (networks/build_backbone.py:265-288, networks/util.py:74-80).
    ``W``, ``b``: the three heads' weights / biases.  -> f [B, 3F] (differentiable torch graph)."""
    x = [torch.cat([m.mean(dim=(2, 3)) for m in maps1], 1), torch.cat([m.mean(dim=(2, 3)) for m in maps2], 1),
         feat3.mean(1)]
    out = []
    for xi, Wi, bi in zip(x, W, b):
        out.append(F.normalize(F.linear(xi, Wi, bi), p=2, dim=1))        # Normalize(2), networks/util.py:74-80
    return torch.cat(out, 1)


# --------------------------------------------------------------------------- #
# row 5 -- dense intra-sample soft InfoNCE     learning/contrast_trainer.py:642-723
# --------------------------------------------------------------------------- #
def dense_keep(depth_mask, h, w):
    """Images kept by ``valid_depth_prob.sum(-1) > 0`` (:677-682) and their
    resized masks (the multinomial weights)."""
    m = nearest_resize_mask(depth_mask, h, w).reshape(depth_mask.shape[0], h * w)
    return m.sum(-1) > 0, m


def dense_soft_nce(map1, map2, sample_ind, keep, temperature, use_depth=None):
    """Row 5 given the sampled pixel indices ``sample_ind [B',S]`` of the kept images.

    A = F2^T F1 / tau (A[i,j] = depth sample i . rgb sample j);
    Tgt = softmax_i(-|q_i - q_j|_2) computed in fp32 (:702-706);
    loss_r2d = -mean_{b,j} sum_i Tgt[i,j] logsoftmax_i(A)[i,j]
    loss_d2r = -mean_{b,i} sum_j Tgt[j,i] logsoftmax_j(A)[i,j]       
    Returns losses[2], accs[2], grad_map1, grad_map2."""
    B, C, h, w = map1.shape
    dt = map1.dtype
    if use_depth is not None and torch.as_tensor(use_depth).sum() == 0:      # :663-665
        return torch.zeros(2, dtype=dt), torch.zeros(2, dtype=dt), torch.zeros_like(map1), torch.zeros_like(map2)
    kept = torch.nonzero(keep).flatten()
    Bk, S = sample_ind.shape
    x1 = gather_pixels(map1[kept], sample_ind)      # [B',S,C]
    x2 = gather_pixels(map2[kept], sample_ind)
    f1 = F.normalize(x1, dim=-1)
    f2 = F.normalize(x2, dim=-1)
    A = torch.bmm(f2, f1.transpose(1, 2)) / temperature          # [B',S(i),S(j)]
    q = torch.stack([sample_ind // w, sample_ind % w], -1).float()
    dist = torch.sqrt(((q[:, :, None, :] - q[:, None, :, :]) ** 2).sum(-1))
    tgt = torch.softmax(-dist, 1).to(dt)                         # column-normalised over i
    ls_col = torch.log_softmax(A, dim=1)
    ls_row = torch.log_softmax(A, dim=2)
    loss_r2d = -(tgt * ls_col).sum(1).mean()
    loss_d2r = -(tgt.transpose(1, 2) * ls_row).sum(2).mean()
    ar = torch.arange(S)
    acc_r2d = (A.argmax(1) == ar).sum(-1).float() / S
    acc_d2r = (A.argmax(2) == ar).sum(-1).float() / S
    G = ((ls_col.exp() - tgt) + (ls_row.exp() - tgt.transpose(1, 2))) / (Bk * S)
    df2 = torch.bmm(G, f1) / temperature
    df1 = torch.bmm(G.transpose(1, 2), f2) / temperature
    dx1 = _normalize_bwd(df1, x1)
    dx2 = _normalize_bwd(df2, x2)
    g1 = torch.zeros_like(map1)
    g2 = torch.zeros_like(map2)
    g1[kept] = scatter_pixels(dx1, sample_ind, (Bk, C, h, w))
    g2[kept] = scatter_pixels(dx2, sample_ind, (Bk, C, h, w))
    return (torch.stack([loss_r2d, loss_d2r]), torch.stack([acc_r2d.mean(), acc_d2r.mean()]).to(dt), g1, g2)


# --------------------------------------------------------------------------- #
# row 6 -- joint <-> graph-node InfoNCE        learning/contrast_trainer.py:744-828
# --------------------------------------------------------------------------- #
def joint_nce(map1, map2, feat3, joints2d, joints_vis, temperature, use_depth=None):
    """A_m[i,j] = Ghat[i] . Fhat_m[:,j] / tau (i: graph node, j: pixel-joint);
    CE over i with target j, ignoring invisible joints (and, for depth, images
    with use_depth==0).  An all-ignored target gives NaN like ``nn.CrossEntropyLoss``.
    Returns losses[2], accs[2], grad_map1, grad_map2, grad_feat3."""
    B, C, h, w = map1.shape
    assert h == w
    dt = map1.dtype
    J = joints_vis.shape[1]
    pix = joint_pixels(joints2d, h)
    x = [gather_pixels(map1, pix), gather_pixels(map2, pix)]        # [B,J,C]
    f = [F.normalize(v, dim=-1) for v in x]
    gh = F.normalize(feat3, dim=-1)
    vis = torch.as_tensor(joints_vis) != 0
    valid = [vis, vis if use_depth is None else vis & (torch.as_tensor(use_depth) != 0)[:, None]]
    losses, accs, dxs = [], [], []
    dgh = torch.zeros_like(gh)
    ar = torch.arange(J)
    for m in range(2):
        A = torch.bmm(gh, f[m].transpose(1, 2)) / temperature        # [B,i,j]
        ls = torch.log_softmax(A, dim=1)
        v = valid[m]
        cnt = v.sum()
        diag = ls[:, ar, ar]                                          # [B,j]
        if cnt == 0:
            losses.append(torch.tensor(float('nan'), dtype=dt))
            dA = torch.zeros_like(A)        # nll_loss backward: ignored targets get 0, so do all of them
        else:
            losses.append(-(diag * v).sum() / cnt)
            onehot = torch.eye(J, dtype=dt).expand(B, J, J)
            dA = (ls.exp() - onehot) * v[:, None, :].to(dt) / cnt
        pred_ok = (A.argmax(1) == ar) & v
        denom = v.sum(-1)
        acc_img = pred_ok.sum(-1).float() / denom.clamp_min(1)
        keep = denom != 0
        accs.append(acc_img[keep].mean() if keep.any() else torch.tensor(float('nan')))
        dgh = dgh + torch.bmm(dA, f[m]) / temperature
        df = torch.bmm(dA.transpose(1, 2), gh) / temperature
        dxs.append(_normalize_bwd(df, x[m]))
    g1 = scatter_pixels(dxs[0], pix, map1.shape)
    g2 = scatter_pixels(dxs[1], pix, map2.shape)
    g3 = _normalize_bwd(dgh, feat3)
    return torch.stack(losses), torch.stack(accs).to(dt), g1, g2, g3


# --------------------------------------------------------------------------- #
# row 7 -- cross-subject SCL                   learning/contrast_trainer.py:830-892
#          (use_rgb=None semantics: learning/segment_trainer.py:601-606)
# --------------------------------------------------------------------------- #
def scl(map1, map2, joints2d, temperature, use_depth, use_rgb=None):
    """X = cat(norm rgb joints, norm depth joints) [2BJ,C]; A = X X^T / tau;
    loss = mean_u( -sum_v pos[u,v] logsoftmax_v(A)[u,v] / max(1, sum_v pos[u,v]) ),
    pos = same joint id, u != v, both endpoints' modality present.
    Returns (loss, grad_map1, grad_map2, early_out)."""
    B, C, h, w = map1.shape
    dt = map1.dtype
    ud = torch.as_tensor(use_depth)
    if ud.sum() == 0:                                                 # :845-847
        return torch.zeros((), dtype=dt), torch.zeros_like(map1), torch.zeros_like(map2), True
    J = joints2d.shape[-2]
    pix = joint_pixels(joints2d, h)
    x = torch.cat([gather_pixels(map1, pix), gather_pixels(map2, pix)], 0)   # [2B,J,C]
    X = F.normalize(x, dim=-1).reshape(2 * B * J, C)
    N = X.shape[0]
    A = X @ X.t() / temperature
    lp = torch.log_softmax(A, dim=1)
    jid = torch.arange(N) % J
    ok_rgb = torch.ones(B, dtype=torch.bool) if use_rgb is None else (torch.as_tensor(use_rgb) != 0)
    valid = torch.cat([ok_rgb, ud != 0]).repeat_interleave(J)
    pos = (jid[:, None] == jid[None, :]) & valid[:, None] & valid[None, :]
    pos.fill_diagonal_(False)
    pos = pos.to(dt)
    npos = pos.sum(1)
    c = npos.clamp_min(1)
    loss = (-(pos * lp).sum(1) / c).mean()
    dA = (-pos / c[:, None] + lp.exp() * (npos / c)[:, None]) / N
    dX = (dA + dA.t()) @ X / temperature
    dx = _normalize_bwd(dX.view(2 * B, J, C), x)
    g1 = scatter_pixels(dx[:B], pix, map1.shape)
    g2 = scatter_pixels(dx[B:], pix, map2.shape)
    return loss, g1, g2, False


# --------------------------------------------------------------------------- #
# the whole hot path of one stage-2 step (used by the CPU-baseline leg of bench.py)
# --------------------------------------------------------------------------- #
def stage2_hot_path(banks, idx, xs, T, m, all_xs, all_y, map1, map2, feat3, sample_ind, keep,
                    joints2d, joints_vis, use_depth, use_rgb, temperature):
    """Rows 2-7 in the order of ``_train_bank_joints_pri3d_cmc3``
    (contrast_trainer.py:954-980).  Returns (total loss, new banks)."""
    losses, accs, gx, _ = bank_nce(banks, idx, xs, T, use_depth=use_depth)
    new_banks = [bank_update(b, ax, all_y, m) for b, ax in zip(banks, all_xs)]
    ld, _, _, _ = dense_soft_nce(map1, map2, sample_ind, keep, temperature, use_depth)
    lj, _, _, _, _ = joint_nce(map1, map2, feat3, joints2d, joints_vis, temperature, use_depth)
    ls, _, _, _ = scl(map1, map2, joints2d, temperature, use_depth, use_rgb)
    return losses.sum() + ld.sum() + lj.sum() + ls, new_banks
