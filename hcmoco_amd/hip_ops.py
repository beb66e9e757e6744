"""torch-facing wrappers over the C ABI of libhcmoco_hip.so.

Every function here takes ROCm tensors, hands raw device pointers + the current HIP stream to
the C entry points of ``include/hcmoco_hip.h`` and wraps the result in a
``torch.autograd.Function`` where a backward exists.  No op has a CPU or eager fallback:
CPU tensors raise ``RuntimeError`` and a missing library raises ``ImportError``.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import Strides4, check


# --------------------------------------------------------------------------- #
# plumbing
# --------------------------------------------------------------------------- #
def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('hcmoco_amd.%s needs ROCm device tensors (no CPU fallback exists)' % name)
    if t.dtype != dtype:
        raise TypeError('%s: expected %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s: tensor must be contiguous' % name)
    p = C.c_void_p(t.data_ptr())
    # The pointer object owns a reference to the tensor until the foreign call has been issued: a temporary
    # (``x.contiguous()``, ``mask.to(int32)``) written inline in an argument list would otherwise be freed
    # before the NEXT argument is evaluated, and the caching allocator hands the same block to the next
    # temporary -- three chunk views of one [B, 384] feature matrix then arrive as three aliases of the last one.
    p._keepalive = t
    return p


def _opt(t, dtype, name):
    return C.c_void_p(0) if t is None else _dev(t, dtype, name)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i32(t):
    """Mask vectors arrive as int64/int32/bool/float from the loader; the ABI takes int32."""
    if t is None:
        return None
    return t.to(dtype=torch.int32).contiguous()


def _bank_entry(name, banks):
    """(C function, torch dtype) for fp32 or bf16 bank storage (BASELINE config 5)."""
    dt = banks[0].dtype
    if dt == torch.float32:
        return getattr(_lib.lib(), name), dt
    if dt == torch.bfloat16:
        return getattr(_lib.lib(), name + '_bf16'), dt
    raise TypeError('banks must be float32 or bfloat16, got %s' % dt)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------- #
# row 1: alias sampler
# --------------------------------------------------------------------------- #
def alias_build(probs):
    """Host-side Walker tables (memory/alias_multinomial.py:7-42) -> (prob fp32[n], alias int64[n])."""
    probs = torch.as_tensor(probs, dtype=torch.float32).clone().contiguous()
    if probs.sum() > 1:
        probs = probs / probs.sum()
    n = probs.numel()
    prob = torch.empty(n, dtype=torch.float32)
    alias = torch.empty(n, dtype=torch.int64)
    check(_lib.lib().hcm_alias_build(C.c_void_p(probs.data_ptr()), n, C.c_void_p(prob.data_ptr()),
                                     C.c_void_p(alias.data_ptr())), 'hcm_alias_build')
    return prob, alias


def alias_draw(prob, alias, y, B, K1, seed, offset, oob=None):
    """idx [B,K1] int64, idx[:,0]=y (mem_bank.py:176-177), Philox-keyed by (seed, offset).  ``oob`` (int32[1] device
    flag): y is clamped into [0, n) and the flag records that a clamp changed something."""
    idx = torch.empty(B, K1, dtype=torch.int64, device=prob.device)
    if oob is not None:
        check(_lib.lib().hcm_alias_draw_checked(
            _dev(prob, torch.float32, 'alias_draw'), _dev(alias, torch.int64, 'alias_draw'), prob.numel(),
            _opt(y, torch.int64, 'alias_draw'), B, K1, seed & (2 ** 64 - 1), offset & (2 ** 64 - 1),
            _dev(idx, torch.int64, 'alias_draw'), _dev(oob, torch.int32, 'alias_draw'), _stream()),
            'hcm_alias_draw_checked')
        return idx
    check(_lib.lib().hcm_alias_draw(_dev(prob, torch.float32, 'alias_draw'), _dev(alias, torch.int64, 'alias_draw'),
                                    prob.numel(), _opt(y, torch.int64, 'alias_draw'), B, K1,
                                    seed & (2 ** 64 - 1), offset & (2 ** 64 - 1),
                                    _dev(idx, torch.int64, 'alias_draw'), _stream()), 'hcm_alias_draw')
    return idx


# --------------------------------------------------------------------------- #
# rows 2+4: fused bank NCE
# --------------------------------------------------------------------------- #
def bank_nce_fused_raw(banks, idx, xs, T, use_depth=None, use_rgb=None, stacked=False):
    """One launch sequence -> (losses[6], accs[6], [gx1,gx2,gx3]); gx = d sum(losses)/dx (``stacked``: one [3,B,D] tensor)."""
    x1, x2, x3 = xs
    B, D = x1.shape
    K1 = idx.shape[1]
    dev = x1.device
    out = torch.empty(12, dtype=torch.float32, device=dev)
    gx = torch.empty(3, B, D, dtype=torch.float32, device=dev)
    ud, ur = _i32(use_depth), _i32(use_rgb)
    L = _lib.lib()
    fn, bdt = _bank_entry('hcm_bank_nce_fused', banks)
    nbytes = L.hcm_bank_nce_workspace_bytes(B, K1, D)
    ws = _ws(nbytes, dev)
    check(fn(
        _dev(banks[0], bdt, 'bank_nce'), _dev(banks[1], bdt, 'bank_nce'),
        _dev(banks[2], bdt, 'bank_nce'), banks[0].shape[0],
        _dev(idx, torch.int64, 'bank_nce'),
        _dev(x1, torch.float32, 'bank_nce'), _dev(x2, torch.float32, 'bank_nce'), _dev(x3, torch.float32, 'bank_nce'),
        _opt(ud, torch.int32, 'bank_nce'), _opt(ur, torch.int32, 'bank_nce'),
        B, K1, D, float(T),
        C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 24),
        C.c_void_p(gx[0].data_ptr()), C.c_void_p(gx[1].data_ptr()), C.c_void_p(gx[2].data_ptr()),
        C.c_void_p(ws.data_ptr()), nbytes, _stream()), 'hcm_bank_nce_fused')
    if stacked:
        return out[:6], out[6:], gx
    return out[:6], out[6:], [gx[0], gx[1], gx[2]]


class _BankNCEFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, x3, bank1, bank2, bank3, idx, T, use_depth, use_rgb):
        losses, accs, gx = bank_nce_fused_raw([bank1, bank2, bank3], idx,
                                              [x1.contiguous(), x2.contiguous(), x3.contiguous()],
                                              T, use_depth, use_rgb)
        ctx.save_for_backward(*gx)
        # the six losses are METERS: marked non-differentiable so that the copies the trainer's running averages hold across
        # steps carry no grad_fn -- a meter with a grad_fn keeps this node, and through it the whole step's graph down to the
        # parameters' AccumulateGrad nodes, alive into the next step (r05: "AccumulateGrad node's stream does not match"
        # in the stage-1 loop: the nodes of the quiet first step were re-used when a side stream produced the gradient)
        meters = losses.clone()
        ctx.mark_non_differentiable(meters, accs)
        return losses.sum(), meters, accs

    @staticmethod
    def backward(ctx, g_total, g_losses, g_accs):
        gx1, gx2, gx3 = ctx.saved_tensors
        return (gx1 * g_total, gx2 * g_total, gx3 * g_total) + (None,) * 7


def bank_nce_fused(xs, banks, idx, T, use_depth=None, use_rgb=None):
    """Differentiable total = sum of the six bank CE losses, plus (detached) losses[6], accs[6]."""
    return _BankNCEFused.apply(xs[0], xs[1], xs[2], banks[0], banks[1], banks[2], idx, T, use_depth, use_rgb)


def bank_nce_fused_timed(banks, idx, xs, T, reps, use_depth=None):
    """Mean ms per fused pass measured with hipEvents on the launch stream (bench/roofline)."""
    x1, x2, x3 = xs
    B, D = x1.shape
    K1 = idx.shape[1]
    dev = x1.device
    out = torch.empty(12, dtype=torch.float32, device=dev)
    gx = torch.empty(3, B, D, dtype=torch.float32, device=dev)
    ud = _i32(use_depth)
    L = _lib.lib()
    nbytes = L.hcm_bank_nce_workspace_bytes(B, K1, D)
    ws = _ws(nbytes, dev)
    ms = C.c_float(0.0)
    fn, bdt = _bank_entry('hcm_bank_nce_fused_timed', banks)
    check(fn(
        _dev(banks[0], bdt, 'bank_nce'), _dev(banks[1], bdt, 'bank_nce'),
        _dev(banks[2], bdt, 'bank_nce'), banks[0].shape[0], _dev(idx, torch.int64, 'bank_nce'),
        _dev(x1, torch.float32, 'bank_nce'), _dev(x2, torch.float32, 'bank_nce'), _dev(x3, torch.float32, 'bank_nce'),
        _opt(ud, torch.int32, 'bank_nce'), C.c_void_p(0), B, K1, D, float(T),
        C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 24),
        C.c_void_p(gx[0].data_ptr()), C.c_void_p(gx[1].data_ptr()), C.c_void_p(gx[2].data_ptr()),
        C.c_void_p(ws.data_ptr()), nbytes, _stream(), int(reps), C.byref(ms)), 'hcm_bank_nce_fused_timed')
    return float(ms.value)


def prof_enable(on):
    """Start/stop (and clear) hipEvent timing of the fused gather pass inside the library."""
    check(_lib.lib().hcm_prof_enable(1 if on else 0), 'hcm_prof_enable')


PROF_TAGS = {'bank_pass': 0, 'dense_stats': 1, 'dense_grad': 2, 'scl_stats': 3, 'scl_grad': 4, 'sgc_fwd': 5,
             'sgc_bwd': 6, 'row8_fwd': 7, 'row8_dw': 8, 'row8_bwd': 9, 'joint': 10,
             'conv1x1_fwd': 11, 'conv1x1_dx': 12, 'conv1x1_dw': 13, 'ball_fwd': 14, 'ball_bwd': 15, 'ballmax_fwd': 16,
             'ballmax_bwd': 17, 'fps': 18, 'three_nn': 19, 'ball_query': 20, 'row8_nhwc': 21}


def prof_read(tag='bank_pass'):
    """(total ms, launches) of the tagged kernel(s) since prof_enable(True); synchronises."""
    total, n = C.c_double(0.0), C.c_int64(0)
    check(_lib.lib().hcm_prof_read_tag(PROF_TAGS[tag], C.byref(total), C.byref(n)), 'hcm_prof_read_tag')
    return float(total.value), int(n.value)


def prof_read_work(tag):
    """Algorithmic work (flops, bytes or distance evaluations, see include/hcmoco_hip.h) the tagged launches stated."""
    w = C.c_double(0.0)
    check(_lib.lib().hcm_prof_read_work(PROF_TAGS[tag], C.byref(w)), 'hcm_prof_read_work')
    return float(w.value)


# --------------------------------------------------------------------------- #
# API mode: materialised logits (the literal CMCMem3.forward contract)
# --------------------------------------------------------------------------- #
class _BankLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, x3, bank1, bank2, bank3, idx, T):
        x1, x2, x3 = x1.contiguous(), x2.contiguous(), x3.contiguous()
        B, D = x1.shape
        K1 = idx.shape[1]
        logits = torch.empty(6, B, K1, dtype=torch.float32, device=x1.device)
        fn, bdt = _bank_entry('hcm_bank_logits_fwd', [bank1])
        check(fn(
            _dev(bank1, bdt, 'bank_logits'), _dev(bank2, bdt, 'bank_logits'),
            _dev(bank3, bdt, 'bank_logits'), bank1.shape[0], _dev(idx, torch.int64, 'bank_logits'),
            _dev(x1, torch.float32, 'bank_logits'), _dev(x2, torch.float32, 'bank_logits'),
            _dev(x3, torch.float32, 'bank_logits'), B, K1, D, float(T),
            C.c_void_p(logits.data_ptr()), _stream()), 'hcm_bank_logits_fwd')
        # the banks are mutated right after forward (mem_bank.py:195-203); backward must see the
        # rows the logits were computed from, so keep a copy of exactly those rows' owners.
        ctx.save_for_backward(bank1.clone(), bank2.clone(), bank3.clone(), idx)
        ctx.T, ctx.shape = float(T), (B, K1, D)
        return logits

    @staticmethod
    def backward(ctx, g):
        bank1, bank2, bank3, idx = ctx.saved_tensors
        B, K1, D = ctx.shape
        g = g.contiguous()
        gx = torch.empty(3, B, D, dtype=torch.float32, device=g.device)
        L = _lib.lib()
        nbytes = L.hcm_bank_nce_workspace_bytes(B, K1, D)
        ws = _ws(nbytes, g.device)
        fn, bdt = _bank_entry('hcm_bank_logits_bwd', [bank1])
        check(fn(
            _dev(bank1, bdt, 'bank_logits'), _dev(bank2, bdt, 'bank_logits'),
            _dev(bank3, bdt, 'bank_logits'), bank1.shape[0], _dev(idx, torch.int64, 'bank_logits'),
            _dev(g, torch.float32, 'bank_logits'), B, K1, D, ctx.T,
            C.c_void_p(gx[0].data_ptr()), C.c_void_p(gx[1].data_ptr()), C.c_void_p(gx[2].data_ptr()),
            C.c_void_p(ws.data_ptr()), nbytes, _stream()), 'hcm_bank_logits_bwd')
        return gx[0], gx[1], gx[2], None, None, None, None, None


def bank_logits(xs, banks, idx, T):
    """logits [6,B,K+1] in the order 12,21,23,32,13,31 (mem_bank.py:186-191), differentiable in x."""
    return _BankLogits.apply(xs[0], xs[1], xs[2], banks[0], banks[1], banks[2], idx, T)


# --------------------------------------------------------------------------- #
# row 3: momentum update (in place, no grad)
# --------------------------------------------------------------------------- #
def _strided_rows(t, ldx, name):
    """Pointer of a [BW, D] fp32 block whose rows are ``ldx`` floats apart (a column slice of a wider matrix)."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError('hcmoco_amd.%s needs fp32 ROCm device tensors (no CPU fallback exists)' % name)
    if t.dim() != 2 or t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) != ldx):
        raise ValueError('%s: expected unit column stride and row stride %d' % (name, ldx))
    p = C.c_void_p(t.data_ptr())
    p._keepalive = t
    return p


@torch.no_grad()
def bank_update(banks, all_xs, all_y, momentum, ldx=None, oob=None):
    BW, D = all_xs[0].shape
    if oob is not None:
        fn, bdt = _bank_entry('hcm_bank_update_checked', banks)
        ldx = D if ldx is None else int(ldx)
        check(fn(
            _dev(banks[0], bdt, 'bank_update'), _dev(banks[1], bdt, 'bank_update'),
            _dev(banks[2], bdt, 'bank_update'), banks[0].shape[0],
            _strided_rows(all_xs[0].detach(), ldx, 'bank_update'), _strided_rows(all_xs[1].detach(), ldx, 'bank_update'),
            _strided_rows(all_xs[2].detach(), ldx, 'bank_update'), ldx,
            _dev(all_y.contiguous(), torch.int64, 'bank_update'), BW, D, float(momentum),
            _dev(oob, torch.int32, 'bank_update'), _stream()), 'hcm_bank_update_checked')
        return
    fn, bdt = _bank_entry('hcm_bank_update', banks)
    check(fn(
        _dev(banks[0], bdt, 'bank_update'), _dev(banks[1], bdt, 'bank_update'),
        _dev(banks[2], bdt, 'bank_update'), banks[0].shape[0],
        _dev(all_xs[0].detach().contiguous(), torch.float32, 'bank_update'),
        _dev(all_xs[1].detach().contiguous(), torch.float32, 'bank_update'),
        _dev(all_xs[2].detach().contiguous(), torch.float32, 'bank_update'),
        _dev(all_y.contiguous(), torch.int64, 'bank_update'), BW, D, float(momentum), _stream()),
        'hcm_bank_update')


# --------------------------------------------------------------------------- #
# MoCo queue (secondary)
# --------------------------------------------------------------------------- #
@torch.no_grad()
def moco_logits(q, k, queue, T):
    B, D = q.shape
    K = queue.shape[0]
    out = torch.empty(B, K + 1, dtype=torch.float32, device=q.device)
    check(_lib.lib().hcm_moco_logits(_dev(q.contiguous(), torch.float32, 'moco'), _dev(k.contiguous(), torch.float32, 'moco'),
                                     _dev(queue, torch.float32, 'moco'), B, K, D, float(T),
                                     _dev(out, torch.float32, 'moco'), _stream()), 'hcm_moco_logits')
    return out


@torch.no_grad()
def moco_enqueue(queue, all_k, index):
    n_new, D = all_k.shape
    K = queue.shape[0]
    check(_lib.lib().hcm_moco_enqueue(_dev(queue, torch.float32, 'moco'),
                                      _dev(all_k.detach().contiguous(), torch.float32, 'moco'),
                                      n_new, K, D, int(index), _stream()), 'hcm_moco_enqueue')
    return (int(index) + n_new) % K


# --------------------------------------------------------------------------- #
# rows 5-7: feature-map losses
# --------------------------------------------------------------------------- #
def _strides(t):
    s = t.stride()
    return Strides4(s[0], s[1], s[2], s[3])


def _check_maps(m1, m2, name):
    if m1.shape != m2.shape or m1.stride() != m2.stride():
        raise ValueError('%s: the two feature maps must share shape and strides' % name)
    if m1.shape[1] != 128:
        raise ValueError('%s: C must be 128' % name)
    if not m1.is_cuda or m1.dtype != torch.float32:
        raise RuntimeError('hcmoco_amd.%s needs fp32 ROCm device tensors (no CPU fallback exists)' % name)


def _dense_map(t):
    """Maps are read in place through their strides; only overlapping/odd layouts are copied."""
    if t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last):
        return t
    return t.contiguous()


def joint_pixels(joints2d, h):
    """pix [B,J] int64 = clamp(floor(j/4),0,h-1) row-major (contrast_trainer.py:757-761)."""
    if joints2d.dtype == torch.float64:
        # the loader yields doubles (datasets/dataset.py:594-596) and the reference floor-divides in
        # that dtype; floor first so the fp32 hand-off cannot move a value across a multiple of 4
        joints2d = torch.floor(joints2d / 4) * 4
    j = joints2d.to(torch.float32).contiguous()
    B, J = j.shape[:2]
    pix = torch.empty(B, J, dtype=torch.int64, device=j.device)
    check(_lib.lib().hcm_joint_pixels(_dev(j, torch.float32, 'joint_pixels'), B * J, h,
                                      _dev(pix, torch.int64, 'joint_pixels'), _stream()), 'hcm_joint_pixels')
    return pix


class _FmapLosses(torch.autograd.Function):
    """Rows 5-7 in one autograd node: the three losses share the two feature maps, so their
    map gradients are accumulated into ONE pair of zero-initialised buffers by the kernels."""

    @staticmethod
    def forward(ctx, map1, map2, feat3, sample_ind, keep, pix, joints_vis, use_depth, use_rgb,
                temperature, do_dense, do_joint, do_scl, coord_ind=None, coord_w=0, gemm_dtype='fp32'):
        map1, map2 = _dense_map(map1), _dense_map(map2)
        _check_maps(map1, map2, 'fmap_losses')
        B, Cc, h, w = map1.shape
        dev = map1.device
        L = _lib.lib()
        st = _strides(map1)
        g1 = torch.zeros_like(map1)        # preserves the memory format -> same strides
        g2 = torch.zeros_like(map2)
        assert g1.stride() == map1.stride()
        out = torch.zeros(9, dtype=torch.float32, device=dev)   # dense4 | joint4 | scl1
        ud, ur = _i32(use_depth), _i32(use_rgb)
        J = pix.shape[1] if pix is not None else 0
        gfeat3 = None
        p1, p2 = C.c_void_p(map1.data_ptr()), C.c_void_p(map2.data_ptr())
        pg1, pg2 = C.c_void_p(g1.data_ptr()), C.c_void_p(g2.data_ptr())
        if do_dense:
            S = sample_ind.shape[1]
            nb = L.hcm_dense_soft_nce_workspace_bytes(B, S, Cc)
            ws = _ws(nb, dev)
            dense = {'bf16': L.hcm_dense_soft_nce_coords_bf16, 'fp32_exact': L.hcm_dense_soft_nce_coords_exact}.get(gemm_dtype, L.hcm_dense_soft_nce_coords)
            check(dense(p1, p2, st, B, Cc, h, w,
                                              _dev(sample_ind, torch.int64, 'dense'),
                                              _opt(coord_ind, torch.int64, 'dense'), int(coord_w),
                                              _dev(_i32(keep), torch.int32, 'dense'),
                                              S, float(temperature), C.c_void_p(out.data_ptr()), pg1, pg2,
                                              C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_dense_soft_nce_coords')
        if do_joint:
            feat3c = feat3.contiguous()
            gfeat3 = torch.empty_like(feat3c)
            nb = L.hcm_joint_nce_workspace_bytes(B, J, Cc)
            ws = _ws(nb, dev)
            check(L.hcm_joint_nce(p1, p2, st, B, Cc, h, w, _dev(feat3c, torch.float32, 'joint'),
                                  _dev(pix, torch.int64, 'joint'), _dev(_i32(joints_vis), torch.int32, 'joint'),
                                  _opt(ud, torch.int32, 'joint'), J, float(temperature),
                                  C.c_void_p(out.data_ptr() + 16), pg1, pg2,
                                  C.c_void_p(gfeat3.data_ptr()), C.c_void_p(ws.data_ptr()), nb, _stream()),
                  'hcm_joint_nce')
        if do_scl:
            nb = L.hcm_scl_workspace_bytes(B, J, Cc)
            ws = _ws(nb, dev)
            scl = {'bf16': L.hcm_scl_bf16, 'fp32_exact': L.hcm_scl_exact}.get(gemm_dtype, L.hcm_scl)
            check(scl(p1, p2, st, B, Cc, h, w, _dev(pix, torch.int64, 'scl'),
                            _dev(ud, torch.int32, 'scl'), _opt(ur, torch.int32, 'scl'), J, float(temperature),
                            C.c_void_p(out.data_ptr() + 32), pg1, pg2,
                            C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_scl')
        ctx.save_for_backward(g1, g2, gfeat3 if gfeat3 is not None else torch.empty(0, device=dev))
        ctx.has_g3 = gfeat3 is not None
        total = out[0] + out[1] + out[4] + out[5] + out[8]
        meters = out.clone()
        ctx.mark_non_differentiable(meters)      # see _BankNCEFused.forward
        return total, meters

    @staticmethod
    def backward(ctx, g_total, g_out):
        g1, g2, g3 = ctx.saved_tensors
        return (g1 * g_total, g2 * g_total, (g3 * g_total) if ctx.has_g3 else None) + (None,) * 13


def fmap_losses(map1, map2, feat3, sample_ind, keep, pix, joints_vis, use_depth, use_rgb, temperature,
                do_dense=True, do_joint=True, do_scl=True, coord_ind=None, coord_w=0, gemm_dtype='fp32'):
    """total (differentiable in map1, map2, feat3) and the 9 detached meters
    [loss_r2d, loss_d2r, acc_r2d, acc_d2r, loss_rgb2j, loss_d2j, acc_rgb2j, acc_d2j, loss_scl].
    ``coord_ind``/``coord_w``: pixel coordinates of the dense soft target when ``sample_ind`` is only
    a gather index (row mode, see ``fmap_losses_rows``).  ``gemm_dtype='bf16'``: the dense and SCL
    contractions on the bf16 matrix cores (BASELINE config 5; fp32 accumulation)."""
    if gemm_dtype not in ('fp32', 'bf16', 'fp32_exact'):
        raise ValueError('gemm_dtype must be fp32 (split-bf16, fp32-accurate), bf16 or fp32_exact')
    return _FmapLosses.apply(map1, map2, feat3, sample_ind, keep, pix, joints_vis, use_depth, use_rgb,
                             temperature, do_dense, do_joint, do_scl, coord_ind, coord_w, gemm_dtype)


# --------------------------------------------------------------------------- #
# row 8, sampled form: project only the pixels the losses read
# --------------------------------------------------------------------------- #
class _SampleRows(torch.autograd.Function):
    """x[b, :, pix[b, r]] sampled bilinearly on the finest grid -> [B, R, C] (hcm_sample_rows);
    backward = hcm_sample_rows_grad (owner-computes scatter, deterministic, when the branch has the
    sampling grid's resolution -- the finest branch, the only use in the trainer; atomics otherwise).
    Used for the finest branch, whose sampling matrix would be too large to materialise."""

    @staticmethod
    def forward(ctx, x, pix, h0, w0):
        x = _dense_map(x)
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hcmoco_amd.sample_rows needs fp32 ROCm tensors (no CPU fallback exists)')
        B, R = pix.shape
        Cc = x.shape[1]
        out = torch.empty(B * R, Cc, dtype=torch.float32, device=x.device)
        check(_lib.lib().hcm_sample_rows(C.c_void_p(x.data_ptr()), _strides(x), B, Cc, x.shape[2], x.shape[3], h0, w0,
                                         _dev(pix, torch.int64, 'sample_rows'), R, C.c_void_p(out.data_ptr()), Cc, 0,
                                         _stream()), 'hcm_sample_rows')
        ctx.save_for_backward(pix)
        ctx.meta = (tuple(x.shape), x.is_contiguous(), h0, w0)
        return out.view(B, R, Cc)

    @staticmethod
    def backward(ctx, g):
        pix, = ctx.saved_tensors
        shape, contig, h0, w0 = ctx.meta
        B, R = pix.shape
        g = g.reshape(B * R, shape[1]).contiguous()
        gx = torch.zeros(shape, dtype=torch.float32, device=g.device)
        if not contig:
            gx = gx.contiguous(memory_format=torch.channels_last)
        check(_lib.lib().hcm_sample_rows_grad(C.c_void_p(g.data_ptr()), shape[1], 0, _strides(gx), B, shape[1], shape[2],
                                              shape[3], h0, w0, _dev(pix, torch.int64, 'sample_rows'), R,
                                              C.c_void_p(gx.data_ptr()), _stream()), 'hcm_sample_rows_grad')
        return gx, None, None, None


class _SampleRowsCoarse(torch.autograd.Function):
    """Bilinear sampling of a COARSE branch at the pixels ``pix`` of the finest grid -> [B, R, C].
    forward: hcm_sample_rows (4 taps per row, ~5 us) -- the dense sampling-matrix GEMM it replaces sat on the
    step's critical path between the encoders' forward and backward (6 x 87 us);
    backward: ``bmm(S^T, g)`` with the dense sampling matrix S (library GEMM: deterministic where a 4-tap
    scatter would need atomics)."""

    @staticmethod
    def forward(ctx, x, pix, S, h0, w0):
        x = _dense_map(x)
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hcmoco_amd.sample_rows needs fp32 ROCm tensors (no CPU fallback exists)')
        B, R = pix.shape
        Cc = x.shape[1]
        out = torch.empty(B * R, Cc, dtype=torch.float32, device=x.device)
        check(_lib.lib().hcm_sample_rows(C.c_void_p(x.data_ptr()), _strides(x), B, Cc, x.shape[2], x.shape[3], h0, w0,
                                         _dev(pix, torch.int64, 'sample_rows'), R, C.c_void_p(out.data_ptr()), Cc, 0,
                                         _stream()), 'hcm_sample_rows')
        ctx.save_for_backward(S)
        ctx.meta = (tuple(x.shape), x.is_contiguous())
        return out.view(B, R, Cc)

    @staticmethod
    def backward(ctx, g):
        S, = ctx.saved_tensors
        shape, contig = ctx.meta
        gx = torch.bmm(S.transpose(1, 2), g.contiguous()).transpose(1, 2).reshape(shape)   # [B, hw, C] -> [B, C, h, w]
        if not contig:
            gx = gx.contiguous(memory_format=torch.channels_last)
        return gx, None, None, None, None


def sampling_matrix(pix, hi, wi, h0, w0):
    """[B, R, hi*wi] dense bilinear sampling matrix of a coarse branch (no grad)."""
    B, R = pix.shape
    S = torch.zeros(B, R, hi * wi, dtype=torch.float32, device=pix.device)
    check(_lib.lib().hcm_sampling_matrix(_dev(pix, torch.int64, 'sampling_matrix'), B * R, hi, wi, h0, w0,
                                         C.c_void_p(S.data_ptr()), _stream()), 'hcm_sampling_matrix')
    return S


def sampled_projection(weight, bias, pix, maps, sampling=None):
    """rows[b, r] = W . [x0[p]; bilinear(x1)[p]; bilinear(x2)[p]; bilinear(x3)[p]] + bias at p = pix[b, r]
    == ``encoder_linear(merge_all_res(maps))[b, :, p]`` (build_backbone.py:243-254) without the
    270-channel concat or the full-resolution projection.  Every branch is sampled by ``hcm_sample_rows``;
    backward: owner-computes scatter for the finest branch, ``bmm`` with the dense sampling matrices for the
    coarse ones (deterministic library GEMMs); the
    projection itself is one ``[B*R, 270] x [270, 128]`` library GEMM.  ``sampling``: matrices from
    ``sampling_matrix`` to share between the two modalities."""
    h0, w0 = maps[0].shape[-2:]
    parts = [_SampleRows.apply(maps[0], pix, h0, w0)]
    for i, m in enumerate(maps[1:]):
        S = sampling[i] if sampling is not None else sampling_matrix(pix, m.shape[2], m.shape[3], h0, w0)
        parts.append(_SampleRowsCoarse.apply(m, pix, S, h0, w0))
    xs = torch.cat(parts, dim=2)
    return torch.nn.functional.linear(xs, weight.reshape(weight.shape[0], -1), bias)


def fmap_losses_rows(rows1, rows2, feat3, S, coord_ind, coord_w, keep, joints_vis, use_depth, use_rgb, temperature,
                     gemm_dtype='fp32'):
    """The three feature-map losses on already-sampled rows [B, S+J, 128] (first S: dense samples,
    last J: joints).  The row matrix is handed to the same kernels as a [B,128,1,S+J] channels-last
    "map" whose pixel index is the row position, so every gather is one 512-byte line."""
    B, R, Cc = rows1.shape
    dev = rows1.device
    m1 = rows1.permute(0, 2, 1).unsqueeze(2)
    m2 = rows2.permute(0, 2, 1).unsqueeze(2)
    ar = torch.arange(R, device=dev, dtype=torch.int64).unsqueeze(0).expand(B, R)
    gather_dense = ar[:, :S].contiguous()
    gather_joint = ar[:, S:].contiguous()
    return fmap_losses(m1, m2, feat3, gather_dense, keep, gather_joint, joints_vis, use_depth, use_rgb, temperature,
                       coord_ind=coord_ind.contiguous(), coord_w=coord_w, gemm_dtype=gemm_dtype)


# --------------------------------------------------------------------------- #
# row 8 helper: bilinear up-sampling used by the HRNet fuse layers and merge_all_res
# --------------------------------------------------------------------------- #
class _UpsampleBilinear(torch.autograd.Function):
    """Forward: hcm_upsample_bilinear2d (coalesced, ~10x faster than ATen's NCHW forward on MI355X).
    Backward: hcm_upsample_bilinear2d_backward (gather form: deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, x, size):
        N, Cc, Hi, Wi = x.shape
        Ho, Wo = int(size[0]), int(size[1])
        ctx.in_shape, ctx.size = (N, Cc, Hi, Wi), (Ho, Wo)
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError('hcmoco_amd.upsample_bilinear needs fp32 ROCm tensors (no CPU fallback exists)')
        nhwc = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if nhwc:        # stay in the encoder's memory format: no layout round trip
            out = torch.empty(N, Cc, Ho, Wo, dtype=torch.float32, device=x.device,
                              memory_format=torch.channels_last)
            check(_lib.lib().hcm_upsample_bilinear2d_nhwc(C.c_void_p(x.data_ptr()), N, Cc, Hi, Wi, Ho, Wo,
                                                          C.c_void_p(out.data_ptr()), _stream()),
                  'hcm_upsample_bilinear2d_nhwc')
            return out
        x = x.contiguous()
        out = torch.empty(N, Cc, Ho, Wo, dtype=torch.float32, device=x.device)
        check(_lib.lib().hcm_upsample_bilinear2d(_dev(x, torch.float32, 'upsample_bilinear'), N * Cc, Hi, Wi, Ho, Wo,
                                                 C.c_void_p(out.data_ptr()), _stream()), 'hcm_upsample_bilinear2d')
        return out

    @staticmethod
    def backward(ctx, g):
        # gather-form kernel (deterministic); a channels-last gradient is brought to NCHW first
        N, Cc, Hi, Wi = ctx.in_shape
        Ho, Wo = ctx.size
        g = g.contiguous()
        gi = torch.empty(N, Cc, Hi, Wi, dtype=torch.float32, device=g.device)
        check(_lib.lib().hcm_upsample_bilinear2d_backward(_dev(g, torch.float32, 'upsample_bilinear'), N * Cc, Hi, Wi,
                                                          Ho, Wo, C.c_void_p(gi.data_ptr()), _stream()),
              'hcm_upsample_bilinear2d_backward')
        return gi, None


def upsample_bilinear(x, size):
    """F.interpolate(x, size=size, mode='bilinear', align_corners=False) for fp32 NCHW ROCm tensors."""
    return _UpsampleBilinear.apply(x, tuple(size))


# --------------------------------------------------------------------------- #
# SemGCN layer: library GEMM + one fused kernel per direction
# --------------------------------------------------------------------------- #
_SGC_WS = {}


def _sgc_ws(B, J, Cc, E):
    key = (B, J, Cc, E)
    n = _SGC_WS.get(key)
    if n is None:
        n = int(_lib.lib().hcm_sgc_workspace_floats(B, J, Cc, E))
        if n == 0:
            raise ValueError('sgc_layer: unsupported shape B=%d J=%d C=%d E=%d (J <= 32, C in {64, 128}, E <= 256)' % key)
        _SGC_WS[key] = n
    return n


class _SgcLayer(torch.autograd.Function):
    """SemGraphConv [+ BatchNorm1d + ReLU] (networks/SGCN/sem_graph_conv.py:34-48, sem_gcn.py:8-28).
    forward: H = X [W0|W1] (rocBLAS) -> hcm_sgc_forward ; backward: hcm_sgc_backward -> two GEMMs."""

    @staticmethod
    def forward(ctx, x, W, e, bias, gamma, beta, running_mean, running_var, graph, has_bn, relu, training,
                momentum, eps):
        B, J, cin = x.shape
        cout = W.shape[2]
        x2 = x.reshape(B * J, cin)
        wcat = W.permute(1, 0, 2).reshape(cin, 2 * cout)            # [W0 | W1]
        H = torch.mm(x2, wcat)
        E = e.numel()
        dev = x.device
        out = torch.empty(B * J, cout, dtype=torch.float32, device=dev)
        xhat = torch.empty(B * J, cout, dtype=torch.float32, device=dev) if has_bn else out
        invstd = torch.empty(cout, dtype=torch.float32, device=dev)
        A = torch.empty(E, dtype=torch.float32, device=dev)
        nul = C.c_void_p(0)
        p = lambda t: nul if t is None else C.c_void_p(t.data_ptr())
        ec = e.reshape(-1).contiguous()
        ws = torch.empty(_sgc_ws(B, J, cout, E), dtype=torch.float32, device=dev)
        check(_lib.lib().hcm_sgc_forward(
            p(H), _dev(ec, torch.float32, 'sgc'), p(graph[0]), p(graph[1]), p(graph[2]), p(graph[3]), p(graph[4]),
            p(bias), p(gamma), p(beta), p(running_mean), p(running_var), B, J, cout, E, int(has_bn), int(relu),
            int(training), float(momentum), float(eps), p(out), p(xhat), p(invstd), p(A), p(ws), _stream()),
            'hcm_sgc_forward')
        ctx.save_for_backward(x2, wcat, H, out, xhat, invstd, A, gamma if gamma is not None else invstd)
        ctx.graph, ctx.dims = graph, (B, J, cin, cout, E, int(has_bn), int(relu), int(training), bias is not None)
        return out.view(B, J, cout)

    @staticmethod
    def backward(ctx, g):
        x2, wcat, H, out, xhat, invstd, A, gamma = ctx.saved_tensors
        B, J, cin, cout, E, has_bn, relu, training, has_bias = ctx.dims
        graph = ctx.graph
        dev = g.device
        g = g.reshape(B * J, cout).contiguous()
        dH = torch.empty(B * J, 2 * cout, dtype=torch.float32, device=dev)
        small = torch.empty(3 * cout + E, dtype=torch.float32, device=dev)
        dgamma, dbeta, dbias, de = small[:cout], small[cout:2 * cout], small[2 * cout:3 * cout], small[3 * cout:]
        p = lambda t: C.c_void_p(t.data_ptr())
        ws = torch.empty(_sgc_ws(B, J, cout, E), dtype=torch.float32, device=dev)
        check(_lib.lib().hcm_sgc_backward(
            p(g), p(out), p(xhat), p(invstd), p(gamma), p(A), p(graph[0]), p(graph[1]), p(graph[2]), p(graph[3]),
            p(graph[4]), p(H), B, J, cout, E, has_bn, relu, training, p(dH), p(dgamma), p(dbeta), p(dbias), p(de),
            p(ws), _stream()), 'hcm_sgc_backward')
        dx = torch.mm(dH, wcat.t()).view(B, J, cin)
        dW = torch.mm(x2.t(), dH).view(cin, 2, cout).permute(1, 0, 2)
        return (dx, dW, de.view(1, E), dbias if has_bias else None, dgamma if has_bn else None,
                dbeta if has_bn else None, None, None, None, None, None, None, None, None)


def sgc_layer(x, W, e, bias, graph, bn=None, relu=False):
    """One SemGCN layer on a ROCm tensor.  ``bn``: an ``nn.BatchNorm1d`` (its running statistics are
    updated in place when it is in training mode) or None; ``graph``: the five int32 index tensors
    of ``networks/sgcn.py:graph_index``."""
    if bn is None:
        return _SgcLayer.apply(x.contiguous(), W, e, bias, None, None, None, None, graph, False, relu, False, 0.0, 1e-5)
    training = bn.training or bn.running_mean is None
    return _SgcLayer.apply(x.contiguous(), W, e, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, graph,
                           True, relu, training, bn.momentum if bn.momentum is not None else 0.1, bn.eps)


# --------------------------------------------------------------------------- #
# encoder normalisation: BatchNorm2d [+ residual] [+ ReLU], two kernels per direction
# --------------------------------------------------------------------------- #
_BN_STATS_FLOATS = {}


def _bn_stats_floats(N, Cc, HW):
    key = (N, Cc, HW)
    n = _BN_STATS_FLOATS.get(key)
    if n is None:
        n = int(_lib.lib().hcm_bn_act_stats_floats(N, Cc, HW))
        if n == 0:
            raise ValueError('bn_act: unsupported shape N=%d C=%d HW=%d (HW must be a multiple of 4)' % key)
        _BN_STATS_FLOATS[key] = n
    return n


def bn_act_supported(x):
    """Shapes/dtypes hcm_bn_act_* covers: fp32 NCHW-contiguous ROCm maps with H*W % 4 == 0."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[1] <= 65535)


def bn_act(x, weight, bias, running_mean, running_var, momentum, eps, residual=None, relu=False):
    """relu?(batch_norm(x; batch statistics) + residual?) with the running statistics updated in place.
    Runs as a C++ autograd node (csrc/torch_glue, torch.ops.hcmoco.bn_act) over hcm_bn_act_*."""
    return _lib.torch_glue().bn_act(x, residual, weight, bias, running_mean, running_var,
                                    float(momentum), float(eps), bool(relu))


# --------------------------------------------------------------------------- #
# weight gradient of the 3x3 / stride 1 / pad 1 encoder convolutions
# --------------------------------------------------------------------------- #
def conv3x3_wgrad(x, dy, ksize=3, stride=1):
    """dW [K,C,k,k] of a bias-free convolution (k = 3 / pad 1 at stride 1 or 2, or k = 1 / pad 0 at stride 1)
    from its input x [N,C,H,W] and the output gradient dy [N,K,H/stride,W/stride]
    (hcm_conv3x3_wgrad / hcm_conv3x3s2_wgrad / hcm_conv1x1_wgrad: fp32 MFMA partial sums + fixed-order
    reduction)."""
    N, Cc, Hx, Wx = x.shape
    K = dy.shape[1]
    if (ksize, stride) not in ((3, 1), (3, 2), (1, 1)):
        raise ValueError('conv3x3_wgrad: kernel size / stride must be (3,1), (3,2) or (1,1)')
    if Hx % stride or Wx % stride or dy.shape != (N, K, Hx // stride, Wx // stride):
        raise ValueError('conv3x3_wgrad: dy must be [N,K,H/stride,W/stride]')
    H, W = Hx // stride, Wx // stride
    name = {(3, 1): 'hcm_conv3x3_wgrad', (3, 2): 'hcm_conv3x3s2_wgrad', (1, 1): 'hcm_conv1x1_wgrad'}[(ksize, stride)]
    nbytes = int(getattr(_lib.lib(), name + '_workspace_bytes')(N, Cc, K, H, W))
    if nbytes == 0:
        raise ValueError('conv3x3_wgrad: unsupported shape (W must be a multiple of 4)')
    ws = _ws(nbytes, x.device)
    dw = torch.empty(K, Cc, ksize, ksize, dtype=torch.float32, device=x.device)
    check(getattr(_lib.lib(), name)(_dev(x, torch.float32, 'conv3x3_wgrad'), _dev(dy, torch.float32, 'conv3x3_wgrad'),
                                    N, Cc, K, H, W, C.c_void_p(dw.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes,
                                    _stream()), name)
    return dw


# --------------------------------------------------------------------------- #
# The loss section of a second-stage step as ONE autograd node (csrc/section.hip):
# heads -> [packed all-gather] -> negative draw -> bank NCE -> bank update -> pixel sampling -> sampled
# merge_all_res + projection -> dense / joint / SCL losses, and in backward: projection, heads, the eight
# branch gradients.  ~35 launches where the module-by-module form needed ~220.
# --------------------------------------------------------------------------- #
def _branches(maps, name):
    """hcm_branches for the four NCHW maps of one HRNet (keeps the tensors alive through the struct).  ``None`` = the
    ABSENT encoder of include/hcmoco_hip.h (HRNetPN: the second modality is not an HRNet)."""
    if maps is None:
        return _lib.Branches()                      # zero-initialised: map[0] == NULL
    if len(maps) != 4:
        raise ValueError('%s: an HRNet returns four branch maps' % name)
    br = _lib.Branches()
    for i, t in enumerate(maps):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError('hcmoco_amd.%s needs fp32 ROCm device tensors (no CPU fallback exists)' % name)
        if t.dim() != 4 or not t.is_contiguous():
            raise ValueError('%s: branch maps must be NCHW-contiguous' % name)
        br.map[i] = t.data_ptr()
        br.C[i], br.H[i], br.W[i] = t.shape[1], t.shape[2], t.shape[3]
    br._keepalive = list(maps)
    return br


def heads_forward(maps1, maps2, feat3, W1, b1, W2, b2, W3, b3, index=None, pooled2=None):
    """(pooled [2,B,Ctot], mean3 [B,D3], ypre [3,B,F], f [B,3F(+2)], fT [3,B,F]) -- build_backbone.py:265-288.
    With ``index`` the rows of ``f`` are the packed rows of the feature/index all-gather.  ``maps2=None`` (absent second
    encoder): ``pooled2 [B, C2]`` is the second head's input, zero-padded to Ctot here (W2 must be [F, Ctot])."""
    B = maps1[0].shape[0]
    Ctot = sum(m.shape[1] for m in maps1)
    J, D3 = feat3.shape[1], feat3.shape[2]
    F = W1.shape[0]
    dev = feat3.device
    ldf = 3 * F + (2 if index is not None else 0)
    pooled = torch.empty(2, B, Ctot, dtype=torch.float32, device=dev)
    if maps2 is None:
        pooled[1].zero_()
        pooled[1, :, :pooled2.shape[1]].copy_(pooled2)
    mean3 = torch.empty(B, D3, dtype=torch.float32, device=dev)
    ypre = torch.empty(3, B, F, dtype=torch.float32, device=dev)
    f = torch.empty(B, ldf, dtype=torch.float32, device=dev)
    fT = torch.empty(3, B, F, dtype=torch.float32, device=dev)
    d = lambda t: _dev(t, torch.float32, 'heads_forward')
    check(_lib.lib().hcm_heads_forward(
        _branches(maps1, 'heads_forward'), _branches(maps2, 'heads_forward'), d(feat3), B, J, Ctot, D3, F,
        d(W1), d(b1), d(W2), d(b2), d(W3), d(b3), _opt(index, torch.int64, 'heads_forward'),
        d(pooled), d(mean3), d(ypre), d(f), ldf, d(fT), _stream()), 'hcm_heads_forward')
    return pooled, mean3, ypre, f, fT


def heads_backward(gfT, scale, pooled, mean3, ypre, W1, W2, W3, gfeat3_joint, J):
    """-> (dW1, db1, dW2, db2, dW3, db3, dpooled [2,B,Ctot], gfeat3 [B,J,D3]); ``scale``: 0-dim device tensor or None."""
    _, B, F = gfT.shape
    Ctot, D3 = pooled.shape[2], mean3.shape[1]
    dev = gfT.device
    dyws = torch.empty(3, B, F, dtype=torch.float32, device=dev)
    dW1, dW2 = torch.empty_like(W1), torch.empty_like(W2)
    dW3 = torch.empty_like(W3)
    db = torch.empty(3, F, dtype=torch.float32, device=dev)
    dpooled = torch.empty(2, B, Ctot, dtype=torch.float32, device=dev)
    gfeat3 = torch.empty(B, J, D3, dtype=torch.float32, device=dev)
    d = lambda t: _dev(t, torch.float32, 'heads_backward')
    check(_lib.lib().hcm_heads_backward(
        d(gfT), _opt(scale, torch.float32, 'heads_backward'), d(pooled), d(mean3), d(ypre), B, J, Ctot, D3, F,
        d(W1), d(W2), d(W3), _opt(gfeat3_joint, torch.float32, 'heads_backward'), d(dyws), d(dW1),
        C.c_void_p(db[0].data_ptr()), d(dW2), C.c_void_p(db[1].data_ptr()), d(dW3), C.c_void_p(db[2].data_ptr()),
        d(dpooled), d(gfeat3), _stream()), 'hcm_heads_backward')
    return dW1, db[0], dW2, db[1], dW3, db[2], dpooled, gfeat3


def pixel_sample(depth_mask, h, w, S, use_depth, joints2d, seed, offset):
    """(pix [B,S+J] int64, coord [B,S] int64, keep [B] int32): contrast_trainer.py:671-685 + :757-761 in one launch,
    drawn by Philox(seed, offset) -- oracle: ``oracle.hcmoco_oracle.pixel_sample_philox``."""
    B, H, W = depth_mask.shape
    J = joints2d.shape[1]
    dev = depth_mask.device
    pix = torch.empty(B, S + J, dtype=torch.int64, device=dev)
    coord = torch.empty(B, S, dtype=torch.int64, device=dev)
    keep = torch.empty(B, dtype=torch.int32, device=dev)
    if joints2d.dtype == torch.float64:      # see joint_pixels(): floor in the loader's dtype first
        joints2d = torch.floor(joints2d / 4) * 4
    check(_lib.lib().hcm_pixel_sample(
        _dev(depth_mask.to(torch.float32).contiguous(), torch.float32, 'pixel_sample'), B, H, W, h, w, S,
        _opt(_i32(use_depth), torch.int32, 'pixel_sample'),
        _dev(joints2d.to(torch.float32).contiguous(), torch.float32, 'pixel_sample'), J,
        seed & (2 ** 64 - 1), offset & (2 ** 64 - 1), _dev(pix, torch.int64, 'pixel_sample'),
        _dev(coord, torch.int64, 'pixel_sample'), _dev(keep, torch.int32, 'pixel_sample'), _stream()),
        'hcm_pixel_sample')
    return pix, coord, keep


def sample_branches(maps1, maps2, pix, Wp1, bp1, Wp2, bp2, zero_grows=True):
    """(xs [2,B*R,ld], Wpad [2,F,ld], grows [2,B*R,F] zero-filled or None): merge_all_res at the pixels ``pix`` for
    both modalities + the packed projections (build_backbone.py:243-254); rows = bmm(xs, Wpad^T)."""
    B, R = pix.shape
    Ctot = sum(m.shape[1] for m in maps1)
    F = Wp1.shape[0]
    ld = int(_lib.lib().hcm_sample_branches_ld(Ctot))
    dev = pix.device
    xs = torch.empty(2, B * R, ld, dtype=torch.float32, device=dev)
    Wpad = torch.empty(2, F, ld, dtype=torch.float32, device=dev)
    grows = torch.empty(2, B * R, F, dtype=torch.float32, device=dev) if zero_grows else None
    d = lambda t: _dev(t, torch.float32, 'sample_branches')
    check(_lib.lib().hcm_sample_branches(
        _branches(maps1, 'sample_branches'), _branches(maps2, 'sample_branches'), B,
        _dev(pix, torch.int64, 'sample_branches'), R, Ctot, F, d(Wp1.reshape(F, Ctot)), d(bp1),
        d(Wp2.reshape(F, Ctot)), d(bp2), d(xs), d(Wpad), _opt(grows, torch.float32, 'sample_branches'), _stream()),
        'hcm_sample_branches')
    return xs, Wpad, grows


def branch_grad(dxs, dpooled, scale, pix, shapes, dWpad=None, F=128, keep=None, S=0):
    """The eight branch gradients (and, from ``dWpad``, the projections' weight/bias gradients) in one launch:
    -> (gmaps1[4], gmaps2[4], dWp1, dbp1, dWp2, dbp2).  ``shapes``: the four branch map shapes [B,C,H,W]."""
    B, R = pix.shape
    dev = pix.device
    Ctot = sum(s[1] for s in shapes)
    g1 = [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    g2 = [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    dWp1 = dbp1 = dWp2 = dbp2 = None
    if dWpad is not None:
        dWp = torch.empty(2, F, Ctot, dtype=torch.float32, device=dev)
        dbp = torch.empty(2, F, dtype=torch.float32, device=dev)
        dWp1, dWp2, dbp1, dbp2 = dWp[0], dWp[1], dbp[0], dbp[1]
    p = lambda t: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
    check(_lib.lib().hcm_branch_grad(
        _opt(dxs, torch.float32, 'branch_grad'), _opt(dpooled, torch.float32, 'branch_grad'),
        _opt(scale, torch.float32, 'branch_grad'), _dev(pix, torch.int64, 'branch_grad'), R, B, Ctot,
        _branches(g1, 'branch_grad'), _branches(g2, 'branch_grad'), _opt(keep, torch.int32, 'branch_grad'), int(S),
        _opt(dWpad, torch.float32, 'branch_grad'), F,
        p(dWp1), p(dbp1), p(dWp2), p(dbp2), _stream()), 'hcm_branch_grad')
    return g1, g2, dWp1, dbp1, dWp2, dbp2


# module attribute: the forward reads channels-last copies of the branches it gathers (hcm_project_rows_cl, r06).  Built,
# bit-identical, measured -- and OFF: project_rows_kernel 99 -> 82.5 us at the bench size, the four copy launches 36.5 us, net
# +20 us per step and no change in samples/s (750.7 / 752.3 on vs 752.3 / 752.1 off, same box, alternating;
# profiles/r06_row8_channels_last.txt).  The producer writes NCHW (the encoder runtime's kernels are per-channel), so the
# coalesced read has to be paid for with a transposing pass that costs more than the scattered one it replaces.
ROW8_CHANNELS_LAST = False


def project_rows(maps1, maps2, pix, Wp1, bp1, Wp2, bp2, save=True, zero_grows=True, channels_last=None):
    """(rows [2,B*R,128], xs [2,B*R,ld] or None, grows [2,B*R,128] zero-filled or None): merge_all_res + the 1x1
    projections (build_backbone.py:243-254, :290-300) at the pixels ``pix`` for both modalities in ONE launch on the fp32
    matrix cores (csrc/rowproj.hip).  ``save``: keep the sampled rows ``xs`` for the weight gradient.
    ``channels_last`` (default ROW8_CHANNELS_LAST = False, see there): hcm_project_rows_cl -- the branch maps that are gathered from global
    memory are first copied [B, H W, C] (one launch), so that a stencil tap is one contiguous run of C floats instead of C
    words on C cache lines; same results bit for bit."""
    B, R = pix.shape
    Ctot = sum(m.shape[1] for m in maps1)
    F = Wp1.shape[0]
    ld = int(_lib.lib().hcm_sample_branches_ld(Ctot))
    dev = pix.device
    rows = torch.empty(2, B * R, F, dtype=torch.float32, device=dev)
    nmod = 1 if maps2 is None else 2          # absent second encoder: the [1] slices are the caller's
    xs = torch.empty(nmod, B * R, ld, dtype=torch.float32, device=dev) if save else None
    grows = torch.empty(2, B * R, F, dtype=torch.float32, device=dev) if zero_grows else None
    if grows is not None and maps2 is None:
        grows[1].zero_()
    d = lambda t: _dev(t, torch.float32, 'project_rows')
    w2 = C.c_void_p(0) if maps2 is None else d(Wp2.reshape(F, Ctot))
    b2 = C.c_void_p(0) if maps2 is None else d(bp2)
    br1, br2 = _branches(maps1, 'project_rows'), _branches(maps2, 'project_rows')
    L = _lib.lib()
    ws, nws = None, 0
    if ROW8_CHANNELS_LAST if channels_last is None else channels_last:
        nws = int(L.hcm_project_rows_nhwc_floats(br1, br2, B, Ctot))
        if nws:
            ws = torch.empty(nws, dtype=torch.float32, device=dev)
    check(L.hcm_project_rows_cl(
        br1, br2, B, _dev(pix, torch.int64, 'project_rows'), R,
        Ctot, F, d(Wp1.reshape(F, Ctot)), d(bp1), w2, b2, _opt(xs, torch.float32, 'project_rows'),
        d(rows), _opt(grows, torch.float32, 'project_rows'), C.c_void_p(0) if ws is None else C.c_void_p(ws.data_ptr()),
        nws, _stream()), 'hcm_project_rows_cl')
    return rows, xs, grows


def project_rows_backward(grows, xs, Wp1, Wp2, dpooled, scale, pix, shapes, keep=None, S=0, weight_grads=True):
    """Backward of ``project_rows`` + the heads' average pooling -> (gmaps1[4], gmaps2[4], dWp1, dbp1, dWp2, dbp2):
    stencil plan, weight-gradient partials + reduction, transposed-order branch tiles (csrc/rowproj.hip)."""
    B, R = pix.shape
    dev = pix.device
    Ctot = sum(s[1] for s in shapes)
    F = grows.shape[-1]
    one = Wp2 is None                           # absent second encoder: modality 0 only
    g1 = [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    g2 = None if one else [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    dWp1 = dbp1 = dWp2 = dbp2 = None
    if weight_grads:
        dWp = torch.empty(1 if one else 2, F, Ctot, dtype=torch.float32, device=dev)
        dbp = torch.empty(1 if one else 2, F, dtype=torch.float32, device=dev)
        dWp1, dbp1 = dWp[0], dbp[0]
        if not one:
            dWp2, dbp2 = dWp[1], dbp[1]
    b1, b2 = _branches(g1, 'project_rows_backward'), _branches(g2, 'project_rows_backward')
    L = _lib.lib()
    nb = L.hcm_project_rows_backward_workspace_bytes(B, R, Ctot, b1)
    ws = _ws(nb, dev)
    p = lambda t: C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())
    d = lambda t: _dev(t, torch.float32, 'project_rows_backward')
    check(L.hcm_project_rows_backward(
        d(grows), d(xs), d(Wp1.reshape(F, Ctot)), C.c_void_p(0) if one else d(Wp2.reshape(F, Ctot)),
        _opt(dpooled, torch.float32, 'project_rows_backward'),
        _opt(scale, torch.float32, 'project_rows_backward'), _dev(pix, torch.int64, 'project_rows_backward'), R, B, Ctot, F,
        b1, b2, _opt(keep, torch.int32, 'project_rows_backward'), int(S), p(dWp1), p(dbp1), p(dWp2), p(dbp2),
        C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_project_rows_backward')
    return g1, g2, dWp1, dbp1, dWp2, dbp2


_ROW_INDEX = {}
GATHER_WAIT_EVENTS = None       # bench.py sets a list: HIP event pairs around the wait for the feature/index all-gather


def _row_index(B, S, J, device):
    """Constant gather indices of the row-matrix 'map' ([B,128,1,S+J] channels-last): dense rows 0..S-1, joints S.."""
    key = (B, S, J, str(device))
    v = _ROW_INDEX.get(key)
    if v is None:
        ar = torch.arange(S + J, device=device, dtype=torch.int64).unsqueeze(0).expand(B, S + J)
        v = _ROW_INDEX[key] = (ar[:, :S].contiguous(), ar[:, S:].contiguous())
    return v


def fmap_losses_on_rows(rows, grows, feat3, S, coord, w, keep, joints_vis, ud, ur, temperature, gemm_dtype='fp32'):
    """Rows 5-7 on the projected row matrix rows [2, B, S+J, 128] (modality major); the row gradients are accumulated
    into the zero-filled ``grows`` (same layout).  -> (meters[9], gfeat3_joint [B,J,128]).  No autograd here: the
    caller (``_Stage2Section``) owns the graph."""
    _, B, R, Cc = rows.shape
    J = R - S
    dev = rows.device
    L = _lib.lib()
    st = Strides4(R * Cc, 1, R * Cc, Cc)
    gd, gj = _row_index(B, S, J, dev)
    out = torch.empty(9, dtype=torch.float32, device=dev)      # dense writes [0:4], joint [4:8], SCL [8]
    p1, p2 = C.c_void_p(rows[0].data_ptr()), C.c_void_p(rows[1].data_ptr())
    pg1, pg2 = C.c_void_p(grows[0].data_ptr()), C.c_void_p(grows[1].data_ptr())
    nb = L.hcm_dense_soft_nce_workspace_bytes(B, S, Cc)
    ws = _ws(nb, dev)
    dense = {'bf16': L.hcm_dense_soft_nce_coords_bf16, 'fp32_exact': L.hcm_dense_soft_nce_coords_exact}.get(gemm_dtype, L.hcm_dense_soft_nce_coords)
    check(dense(p1, p2, st, B, Cc, 1, R, _dev(gd, torch.int64, 'dense'), _dev(coord, torch.int64, 'dense'), int(w),
                _dev(keep, torch.int32, 'dense'), S, float(temperature), C.c_void_p(out.data_ptr()), pg1, pg2,
                C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_dense_soft_nce_coords')
    feat3c = feat3.contiguous()
    gfeat3 = torch.empty_like(feat3c)
    nb = L.hcm_joint_nce_workspace_bytes(B, J, Cc)
    ws = _ws(nb, dev)
    check(L.hcm_joint_nce(p1, p2, st, B, Cc, 1, R, _dev(feat3c, torch.float32, 'joint'), _dev(gj, torch.int64, 'joint'),
                          _dev(joints_vis, torch.int32, 'joint'), _opt(ud, torch.int32, 'joint'), J, float(temperature),
                          C.c_void_p(out.data_ptr() + 16), pg1, pg2, C.c_void_p(gfeat3.data_ptr()),
                          C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_joint_nce')
    nb = L.hcm_scl_workspace_bytes(B, J, Cc)
    ws = _ws(nb, dev)
    scl = {'bf16': L.hcm_scl_bf16, 'fp32_exact': L.hcm_scl_exact}.get(gemm_dtype, L.hcm_scl)
    udv = ud if ud is not None else torch.ones(B, dtype=torch.int32, device=dev)
    check(scl(p1, p2, st, B, Cc, 1, R, _dev(gj, torch.int64, 'scl'), _dev(udv, torch.int32, 'scl'),
              _opt(ur, torch.int32, 'scl'), J, float(temperature), C.c_void_p(out.data_ptr() + 32), pg1, pg2,
              C.c_void_p(ws.data_ptr()), nb, _stream()), 'hcm_scl')
    return out, gfeat3


class _Stage2Section(torch.autograd.Function):
    """forward(feat3, W1, b1, W2, b2, W3, b3, Wp1, bp1, Wp2, bp2, *maps1, *maps2, cfg) -> (total, bank_losses[6],
    bank_accs[6], meters[9]); ``cfg``: plain-python description of the step (see ``stage2_section``)."""

    @staticmethod
    def forward(ctx, feat3, W1, b1, W2, b2, W3, b3, Wp1, bp1, Wp2, bp2, m10, m11, m12, m13, m20, m21, m22, m23, cfg):
        maps1, maps2 = [m10, m11, m12, m13], [m20, m21, m22, m23]
        mem, tape = cfg['contrast'], cfg.get('tape')
        index = cfg['index'].contiguous()
        B, J = feat3.shape[0], feat3.shape[1]
        F = W1.shape[0]
        feat3c = feat3.contiguous()
        gather = cfg.get('gather')
        # ---- heads (and the packed all-gather row when there are other ranks to tell)
        pooled, mean3, ypre, f, fT = heads_forward(maps1, maps2, feat3c, W1, b1, W2, b2, W3, b3,
                                                   index if gather is not None else None)
        pending = None
        if gather is not None:
            # the one collective of the forward pass: started here, waited for where its result is first needed -- the
            # bank update, which only has to come after the NCE pass has read the rows.  Nothing in between depends on
            # the other ranks, so a rank that arrives late costs the others nothing until then.
            allp, pending = gather(f)                                     # [B*W, 3F+2], rank-major
            if pending is not None and tape is not None:
                pending.wait()
                pending = None
            all_x = [allp[:, i * F:(i + 1) * F] for i in range(3)]
            all_index_of = lambda: allp[:, 3 * F:].contiguous().view(torch.int64).view(-1)
            ldx = allp.shape[1]
        else:
            all_x, ldx = [fT[0], fT[1], fT[2]], F
            all_index_of = lambda: index
        # ---- rows 1-4: negatives, fused bank NCE (gx = d sum(losses) / d fT), momentum update after the reads
        ud, ur = _i32(cfg.get('use_depth')), _i32(cfg.get('use_rgb'))
        idx = cfg.get('idx')
        idx = mem.draw(index) if idx is None else mem._in_range(idx).contiguous()
        if tape is not None:
            tape.update(banks0=[b.detach().clone() for b in mem.banks()], idx=idx, f=f[:, :3 * F], fT=fT,
                        all_x=[a.detach().clone() for a in all_x], all_index=all_index_of())
        losses, accs, gxT = bank_nce_fused_raw(mem.banks(), idx, [fT[0], fT[1], fT[2]], mem.T, ud,
                                               ur if cfg.get('bank_use_rgb') else None, stacked=True)

        def update_banks():
            if pending is not None:
                if GATHER_WAIT_EVENTS is not None:      # bench.py: how long the stream stands still for the all-gather
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    pending.wait()
                    ev[1].record()
                    GATHER_WAIT_EVENTS.append(ev)
                else:
                    pending.wait()
            mem.update_strided(all_x, ldx, all_index_of())
        if pending is None:
            update_banks()
        meters = None
        ctx.stage2 = bool(cfg.get('stage2', True))
        if ctx.stage2:
            # ---- sampled pixels (device Philox sampler unless the caller injects them)
            h, w = maps1[0].shape[-2:]
            S = int(cfg['num_samples'])
            vis = _i32(cfg['joints_vis'])
            if cfg.get('sample_ind') is not None:
                pj = joint_pixels(cfg['joints2d'], h)
                coord = cfg['sample_ind'].contiguous()
                pix = torch.cat([coord, pj], dim=1).contiguous()
                keep = _i32(cfg['keep'])
            else:
                pix, coord, keep = pixel_sample(cfg['depth_mask'], h, w, S, ud, cfg['joints2d'], *mem.next_pixel_key())
            # ---- row 8 at the sampled pixels: merge + 1x1 projections (bias folded in) in one MFMA launch
            rows, xs, grows = project_rows(maps1, maps2, pix, Wp1, bp1, Wp2, bp2)          # [2, B*R, F]
            R = pix.shape[1]
            meters, gj = fmap_losses_on_rows(rows.view(2, B, R, F), grows.view(2, B, R, F), feat3c, S, coord, w, keep,
                                             vis, ud, ur, cfg['temperature'], cfg.get('gemm_dtype', 'fp32'))
            if tape is not None:
                tape.update(pix=pix, coord=coord, keep=keep, rows=rows, meters=meters)
            ctx.save_for_backward(pooled, mean3, ypre, W1, W2, W3, gxT, gj, xs, Wp1, Wp2, grows, pix, keep)
            ctx.S = S
        else:
            ctx.save_for_backward(pooled, mean3, ypre, W1, W2, W3, gxT)
        if pending is not None:
            update_banks()
        ctx.J = J
        ctx.shapes = [tuple(m.shape) for m in maps1]
        total = torch.empty((), dtype=torch.float32, device=losses.device)
        check(_lib.lib().hcm_section_total(_dev(losses, torch.float32, 'section'), _opt(meters, torch.float32, 'section'),
                                           C.c_void_p(total.data_ptr()), _stream()), 'hcm_section_total')
        outs = (total, losses, accs, meters if meters is not None else total.new_zeros(9))
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_total, *_unused):
        scale = g_total.contiguous().to(torch.float32)
        if ctx.stage2:
            pooled, mean3, ypre, W1, W2, W3, gxT, gj, xs, Wp1, Wp2, grows, pix, keep = ctx.saved_tensors
            F = W1.shape[0]
            dW1, db1, dW2, db2, dW3, db3, dpooled, gfeat3 = heads_backward(gxT, scale, pooled, mean3, ypre, W1, W2, W3,
                                                                           gj, ctx.J)
            g1, g2, dWp1, dbp1, dWp2, dbp2 = project_rows_backward(grows, xs, Wp1, Wp2, dpooled, scale, pix, ctx.shapes,
                                                                   keep, ctx.S)
            Ctot = dWp1.shape[1]
            return (gfeat3, dW1, db1, dW2, db2, dW3, db3, dWp1.view(F, Ctot, 1, 1), dbp1, dWp2.view(F, Ctot, 1, 1), dbp2,
                    g1[0], g1[1], g1[2], g1[3], g2[0], g2[1], g2[2], g2[3], None)
        pooled, mean3, ypre, W1, W2, W3, gxT = ctx.saved_tensors
        dW1, db1, dW2, db2, dW3, db3, dpooled, gfeat3 = heads_backward(gxT, scale, pooled, mean3, ypre, W1, W2, W3,
                                                                       None, ctx.J)
        # stage 1: the branch maps only feed the average pools
        nopix = torch.empty(ctx.shapes[0][0], 0, dtype=torch.int64, device=gxT.device)
        g1, g2, _, _, _, _ = branch_grad(None, dpooled, None, nopix, ctx.shapes)
        return (gfeat3, dW1, db1, dW2, db2, dW3, db3, None, None, None, None,
                g1[0], g1[1], g1[2], g1[3], g2[0], g2[1], g2[2], g2[3], None)


class _Stage2SectionPN(torch.autograd.Function):
    """The loss section of the ``HRNetPN`` model (networks/build_backbone.py:305-514 of the reference; r05) as ONE autograd
    node.  Same kernels as ``_Stage2Section``; the second modality is not an HRNet, so (include/hcmoco_hip.h, "absent second
    encoder") head 2 pools the cloud features ``feat2 [B, C2, Npts]`` (mean over the points, :486) and its rows are a plain
    gather of the already projected depth map ``lm2 [B, F, h, w]`` (:499-505: Conv1d + pts2depth + nearest resize run in the
    model).  forward(feat3, W1, b1, W2, b2, W3, b3, Wp1, bp1, m10..m13, feat2, lm2, cfg) -> like ``_Stage2Section``."""

    @staticmethod
    def forward(ctx, feat3, W1, b1, W2, b2, W3, b3, Wp1, bp1, m10, m11, m12, m13, feat2, lm2, cfg):
        maps1 = [m10, m11, m12, m13]
        mem, tape = cfg['contrast'], cfg.get('tape')
        index = cfg['index'].contiguous()
        B, J = feat3.shape[0], feat3.shape[1]
        F = W1.shape[0]
        Ctot = sum(m.shape[1] for m in maps1)
        C2 = feat2.shape[1]
        if C2 > Ctot or lm2.shape[1] != F or tuple(lm2.shape[-2:]) != tuple(m10.shape[-2:]):
            raise ValueError('stage2_section_pn: feat2 %s / lm2 %s do not fit the HRNet maps %s'
                             % (tuple(feat2.shape), tuple(lm2.shape), tuple(m10.shape)))
        feat3c = feat3.contiguous()
        gather = cfg.get('gather')
        # ---- heads: head 2's input is the mean over the points, zero-padded to the HRNet's channel count
        pooled2 = feat2.mean(-1)
        W2pad = torch.zeros(F, Ctot, dtype=torch.float32, device=W2.device)
        W2pad[:, :C2].copy_(W2)
        pooled, mean3, ypre, f, fT = heads_forward(maps1, None, feat3c, W1, b1, W2pad, b2, W3, b3,
                                                   index if gather is not None else None, pooled2=pooled2)
        pending = None
        if gather is not None:
            allp, pending = gather(f)
            if pending is not None and tape is not None:
                pending.wait()
                pending = None
            all_x = [allp[:, i * F:(i + 1) * F] for i in range(3)]
            all_index_of = lambda: allp[:, 3 * F:].contiguous().view(torch.int64).view(-1)
            ldx = allp.shape[1]
        else:
            all_x, ldx = [fT[0], fT[1], fT[2]], F
            all_index_of = lambda: index
        ud, ur = _i32(cfg.get('use_depth')), _i32(cfg.get('use_rgb'))
        idx = cfg.get('idx')
        idx = mem.draw(index) if idx is None else mem._in_range(idx).contiguous()
        if tape is not None:
            tape.update(banks0=[b.detach().clone() for b in mem.banks()], idx=idx, f=f[:, :3 * F], fT=fT,
                        all_x=[a.detach().clone() for a in all_x], all_index=all_index_of())
        losses, accs, gxT = bank_nce_fused_raw(mem.banks(), idx, [fT[0], fT[1], fT[2]], mem.T, ud, None, stacked=True)

        def update_banks():
            if pending is not None:
                pending.wait()
            mem.update_strided(all_x, ldx, all_index_of())
        if pending is None:
            update_banks()
        h, w = m10.shape[-2:]
        S = int(cfg['num_samples'])
        vis = _i32(cfg['joints_vis'])
        if cfg.get('sample_ind') is not None:
            pj = joint_pixels(cfg['joints2d'], h)
            coord = cfg['sample_ind'].contiguous()
            pix = torch.cat([coord, pj], dim=1).contiguous()
            keep = _i32(cfg['keep'])
        else:
            pix, coord, keep = pixel_sample(cfg['depth_mask'], h, w, S, ud, cfg['joints2d'], *mem.next_pixel_key())
        R = pix.shape[1]
        # ---- rows: modality 0 = merge + projection at the sampled pixels (MFMA), modality 1 = gather of the depth map
        rows, xs, grows = project_rows(maps1, None, pix, Wp1, bp1, None, None)
        lm2c = _dense_map(lm2)
        check(_lib.lib().hcm_sample_rows(C.c_void_p(lm2c.data_ptr()), _strides(lm2c), B, F, h, w, h, w,
                                         _dev(pix, torch.int64, 'section_pn'), R, C.c_void_p(rows[1].data_ptr()), F, 0,
                                         _stream()), 'hcm_sample_rows')
        meters, gj = fmap_losses_on_rows(rows.view(2, B, R, F), grows.view(2, B, R, F), feat3c, S, coord, w, keep, vis, ud,
                                         ur, cfg['temperature'], cfg.get('gemm_dtype', 'fp32'))
        if tape is not None:
            tape.update(pix=pix, coord=coord, keep=keep, rows=rows, meters=meters)
        ctx.save_for_backward(pooled, mean3, ypre, W1, W2pad, W3, gxT, gj, xs, Wp1, grows, pix, keep)
        ctx.S, ctx.J, ctx.C2 = S, J, C2
        ctx.shapes = [tuple(m.shape) for m in maps1]
        ctx.feat2_shape, ctx.lm2_shape = tuple(feat2.shape), tuple(lm2.shape)
        if pending is not None:
            update_banks()
        total = torch.empty((), dtype=torch.float32, device=losses.device)
        check(_lib.lib().hcm_section_total(_dev(losses, torch.float32, 'section'), _dev(meters, torch.float32, 'section'),
                                           C.c_void_p(total.data_ptr()), _stream()), 'hcm_section_total')
        outs = (total, losses, accs, meters)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_total, *_unused):
        scale = g_total.contiguous().to(torch.float32)
        pooled, mean3, ypre, W1, W2pad, W3, gxT, gj, xs, Wp1, grows, pix, keep = ctx.saved_tensors
        F = W1.shape[0]
        C2 = ctx.C2
        B, R = pix.shape
        dW1, db1, dW2pad, db2, dW3, db3, dpooled, gfeat3 = heads_backward(gxT, scale, pooled, mean3, ypre, W1, W2pad, W3,
                                                                         gj, ctx.J)
        g1, _, dWp1, dbp1, _, _ = project_rows_backward(grows, xs, Wp1, None, dpooled, scale, pix, ctx.shapes, keep, ctx.S)
        # head 2: d mean over the points; depth map: owner-computes scatter of its (scaled) row gradients
        npts = ctx.feat2_shape[2]
        dfeat2 = (dpooled[1, :, :C2] * (1.0 / npts)).unsqueeze(-1).expand(ctx.feat2_shape)
        g2rows = (grows[1] * scale).contiguous()
        dlm2 = torch.zeros(ctx.lm2_shape, dtype=torch.float32, device=g2rows.device)
        h, w = ctx.lm2_shape[-2:]
        check(_lib.lib().hcm_sample_rows_grad(C.c_void_p(g2rows.data_ptr()), F, 0, _strides(dlm2), B, F, h, w, h, w,
                                              _dev(pix, torch.int64, 'section_pn'), R, C.c_void_p(dlm2.data_ptr()),
                                              _stream()), 'hcm_sample_rows_grad')
        Ctot = dWp1.shape[1]
        return (gfeat3, dW1, db1, dW2pad[:, :C2].contiguous(), db2, dW3, db3, dWp1.view(F, Ctot, 1, 1), dbp1,
                g1[0], g1[1], g1[2], g1[3], dfeat2, dlm2, None)


def stage2_section_pn(feat3, heads, proj1, maps1, feat2, lm2, cfg):
    """``stage2_section`` for the HRNetPN model: heads = (W1,b1,W2,b2,W3,b3) with W2 [F, C2] over the cloud features,
    proj1 = (Wp1, bp1) of ``encoder1_linear``; feat2 [B, C2, Npts], lm2 [B, F, h, w] (see ``_Stage2SectionPN``)."""
    return _Stage2SectionPN.apply(feat3, *heads, *proj1, *maps1, feat2, lm2, cfg)


def stage2_section(feat3, heads, projs, maps1, maps2, cfg):
    """One autograd node for the whole loss section.  heads = (W1,b1,W2,b2,W3,b3) of the three Linear heads, projs =
    (Wp1,bp1,Wp2,bp2) of the two 1x1 projections; cfg keys: contrast (CMCMem3), index, use_depth, use_rgb,
    depth_mask, joints2d, joints_vis, num_samples, temperature, gemm_dtype, gather (callable or None), and the
    parity-mode injections idx / sample_ind / keep; ``tape`` (dict) receives the intermediates."""
    return _Stage2Section.apply(feat3, *heads, *projs, *maps1, *maps2, cfg)
