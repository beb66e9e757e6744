#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER.

Imports the *reference* (hongfz16/HCMoCo, mounted read-only at /root/reference)
unmodified, drives its own functions on seeded CPU inputs and writes small
``.npz`` fixtures next to this file.  Nothing of the reference travels: only
inputs / expected outputs are committed.  The GPU box never runs this script.

Shims (process-local, reference files untouched):
  1. ``torch.Tensor.cuda`` / ``torch.nn.Module.cuda`` -> identity
  2. stub modules ``tensorboard_logger``, ``pointnet2_cuda``
  3. a minimal ``yacs.config.CfgNode`` (attr-dict + merge_from_file via PyYAML)
  4. cwd = /root/reference/pycontrast (the reference opens yaml by relative path)

Usage:  python tests/golden/gen_golden.py
"""
import os
import sys
import types
import argparse

import numpy as np
import torch

REF = '/root/reference/pycontrast'
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# shims
# --------------------------------------------------------------------------- #
def install_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for name in ('tensorboard_logger', 'pointnet2_cuda'):
        sys.modules[name] = types.ModuleType(name)

    import yaml

    class CfgNode(dict):
        def __init__(self, init=None, **kw):
            super().__init__()
            for k, v in (init or {}).items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def defrost(self):
            pass

        def freeze(self):
            pass

        def _merge(self, other):
            for k, v in other.items():
                if isinstance(v, dict) and isinstance(self.get(k), dict):
                    self[k]._merge(v)
                else:
                    self[k] = CfgNode(v) if isinstance(v, dict) else v

        def merge_from_file(self, path):
            with open(path) as f:
                self._merge(yaml.safe_load(f))

        def merge_from_list(self, lst):
            pass

    yacs = types.ModuleType('yacs')
    yacs_config = types.ModuleType('yacs.config')
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules['yacs'] = yacs
    sys.modules['yacs.config'] = yacs_config

    sys.path.insert(0, REF)
    os.chdir(REF)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote %-28s %7.1f KiB' % (name + '.npz', os.path.getsize(path) / 1024))


def f32(x):
    return float(np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x).reshape(-1)[0])


# --------------------------------------------------------------------------- #
# 1. alias tables  (memory/alias_multinomial.py)
# --------------------------------------------------------------------------- #
def gen_alias():
    from memory.alias_multinomial import AliasMethod
    g = torch.Generator().manual_seed(11)
    p = torch.rand(97, generator=g) + 0.05
    p_in = p.clone()
    am = AliasMethod(p.clone())
    uni = AliasMethod(torch.ones(1000))
    uni2 = AliasMethod(torch.ones(4096))
    npz('alias_tables', probs=p_in, prob=am.prob, alias=am.alias,
        uni1000_prob=uni.prob, uni1000_alias=uni.alias,
        uni4096_prob=uni2.prob, uni4096_alias=uni2.alias)


# --------------------------------------------------------------------------- #
# 2. bank NCE  (memory/mem_bank.py CMCMem3 + contrast_trainer._compute_loss_accuracy)
# --------------------------------------------------------------------------- #
def gen_bank():
    from memory.mem_bank import CMCMem3
    from learning.contrast_trainer import ContrastTrainer
    import torch.nn as nn

    torch.manual_seed(1234)
    B, D, n, K, W = 6, 128, 256, 64, 2
    T, mom = 0.07, 0.5
    mem = CMCMem3(D, n, K, T, mom)
    bank0 = [mem.memory_1.clone(), mem.memory_2.clone(), mem.memory_3.clone()]

    drawn = {}
    orig_draw = mem.multinomial.draw

    def draw(N):
        out = orig_draw(N)
        drawn['idx'] = out.clone()
        return out
    mem.multinomial.draw = draw

    f = torch.nn.functional.normalize(torch.randn(B * W, 3 * D).view(B * W, 3, D), dim=2)
    all_x = [f[:, i].contiguous() for i in range(3)]
    # global indices: a duplicate inside rank 0, and one across ranks (last wins)
    all_y = torch.tensor([5, 17, 200, 17, 33, 255, 90, 5, 101, 7, 64, 128], dtype=torch.long)
    x = [a[:B].clone().requires_grad_(True) for a in all_x]
    y = all_y[:B].clone()

    out = mem(x[0], x[1], x[2], y, all_x[0], all_x[1], all_x[2], all_y)
    logits, labels = out[:-1], out[-1]
    idx = drawn['idx'].view(B, K + 1).clone()
    idx[:, 0] = y
    bank1 = [mem.memory_1.clone(), mem.memory_2.clone(), mem.memory_3.clone()]

    crit = nn.CrossEntropyLoss()
    regimes = {
        'none':      dict(use_depth=None, use_rgb=None),
        'depth_mix': dict(use_depth=torch.tensor([1, 0, 1, 1, 0, 1]), use_rgb=None),
        'depth_all0': dict(use_depth=torch.zeros(B, dtype=torch.long), use_rgb=None),
        'both_mix':  dict(use_depth=torch.tensor([1, 0, 1, 1, 0, 1]),
                          use_rgb=torch.tensor([1, 1, 0, 1, 1, 1])),
        'both_none': dict(use_depth=torch.tensor([1, 0, 1, 0, 0, 1]),
                          use_rgb=torch.tensor([0, 1, 0, 1, 1, 0])),
    }
    arrays = dict(B=B, D=D, n=n, K=K, T=T, m=mom,
                  bank0_1=bank0[0], bank0_2=bank0[1], bank0_3=bank0[2],
                  bank1_1=bank1[0], bank1_2=bank1[1], bank1_3=bank1[2],
                  idx=idx, y=y, all_y=all_y,
                  x1=x[0], x2=x[1], x3=x[2],
                  all_x1=all_x[0], all_x2=all_x[1], all_x3=all_x[2],
                  labels=labels)
    for i, l in enumerate(logits):
        arrays['logits%d' % i] = l
    for name, reg in regimes.items():
        losses, accs = ContrastTrainer._compute_loss_accuracy(
            logits=list(logits), target=labels, criterion=crit, **reg)
        total = sum(losses)
        grads = torch.autograd.grad(total, x, retain_graph=True, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(x[0]) for g in grads]
        arrays[name + '_losses'] = np.array([f32(l) for l in losses], dtype=np.float32)
        arrays[name + '_accs'] = np.array([f32(a) for a in accs], dtype=np.float32)
        for i, g in enumerate(grads):
            arrays[name + '_gx%d' % (i + 1)] = g
        if reg['use_depth'] is not None:
            arrays[name + '_use_depth'] = reg['use_depth']
        if reg['use_rgb'] is not None:
            arrays[name + '_use_rgb'] = reg['use_rgb']
    npz('bank_nce', **arrays)


# --------------------------------------------------------------------------- #
# 3. MoCo queue (memory/mem_moco.py CMCMoCo) -- pointer bookkeeping over two steps
# --------------------------------------------------------------------------- #
def gen_moco():
    from memory.mem_moco import CMCMoCo
    torch.manual_seed(77)
    D, K, B = 128, 48, 5
    moco = CMCMoCo(D, K, 0.2)
    arrays = dict(D=D, K=K, B=B, T=0.2, queue0_1=moco.memory_1.clone(), queue0_2=moco.memory_2.clone())
    nrm = torch.nn.functional.normalize
    for step in range(3):
        q1, k1, q2, k2 = [nrm(torch.randn(B, D)) for _ in range(4)]
        all_k1 = torch.cat([k1, nrm(torch.randn(B, D))])
        all_k2 = torch.cat([k2, nrm(torch.randn(B, D))])
        l1, l2, lab = moco(q1, k1, q2, k2, all_k1=all_k1, all_k2=all_k2)
        arrays.update({'s%d_q1' % step: q1, 's%d_k1' % step: k1, 's%d_q2' % step: q2,
                       's%d_k2' % step: k2, 's%d_all_k1' % step: all_k1, 's%d_all_k2' % step: all_k2,
                       's%d_logits1' % step: l1, 's%d_logits2' % step: l2,
                       's%d_index' % step: moco.index,
                       's%d_queue_1' % step: moco.memory_1.clone(),
                       's%d_queue_2' % step: moco.memory_2.clone()})
    npz('moco_queue', **arrays)


# --------------------------------------------------------------------------- #
# 4-6. feature-map losses (learning/contrast_trainer.py :642-892)
# --------------------------------------------------------------------------- #
def make_trainer(temperature=0.07, num_samples=16):
    from learning.contrast_trainer import ContrastTrainer
    tr = ContrastTrainer.__new__(ContrastTrainer)
    tr.args = argparse.Namespace(temperature=temperature, pri3d_num_samples_per_image=num_samples)
    return tr


def gen_dense():
    torch.manual_seed(2024)
    B, C, h, S = 5, 128, 8, 24
    H = 4 * h
    tr = make_trainer(num_samples=S)
    m1 = torch.randn(B, C, h, h, requires_grad=True)
    m2 = torch.randn(B, C, h, h, requires_grad=True)
    depth = torch.randn(B, H, H)
    mask = torch.zeros(B, H, H)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing='ij')
    disc = ((yy - H / 2) ** 2 + (xx - H / 2) ** 2) < (0.37 * H) ** 2
    mask[:, disc] = 1.0
    mask[3] = 0.0                       # an image with an empty mask is dropped (:677-682)
    use_depth = torch.tensor([1, 1, 1, 0, 1])

    captured = {}
    orig = torch.Tensor.multinomial

    def capture(self, *a, **k):
        out = orig(self, *a, **k)
        captured['ind'] = out.clone()
        return out
    torch.Tensor.multinomial = capture
    try:
        losses, accs = tr._compute_soft_pri3d_loss_accuracy(
            m1, m2, depth, None, use_depth=use_depth, depth_mask=mask, scale=None)
    finally:
        torch.Tensor.multinomial = orig
    g1, g2 = torch.autograd.grad(sum(losses), [m1, m2])
    # all-zero use_depth early return
    l0, a0 = tr._compute_soft_pri3d_loss_accuracy(
        m1, m2, depth, None, use_depth=torch.zeros(B, dtype=torch.long), depth_mask=mask)
    npz('dense_soft_nce', B=B, C=C, h=h, S=S, temperature=0.07,
        map1=m1, map2=m2, depth_mask=mask, use_depth=use_depth,
        sample_ind=captured['ind'],          # [B', S] indices into h*w for the kept images
        losses=np.array([f32(l) for l in losses], np.float32),
        accs=np.array([f32(a) for a in accs], np.float32),
        grad_map1=g1, grad_map2=g2,
        zero_losses=np.array([f32(l) for l in l0], np.float32))


def gen_joint():
    arrays = {}
    for J in (13, 16, 17):
        torch.manual_seed(300 + J)
        B, C, h = 4, 128, 8
        tr = make_trainer()
        m1 = torch.randn(B, C, h, h, requires_grad=True)
        m2 = torch.randn(B, C, h, h, requires_grad=True)
        g3 = torch.randn(B, J, C, requires_grad=True)
        j2d = torch.rand(B, J, 2) * (4 * h)
        j2d[0, 0] = torch.tensor([-3.0, 5.0])          # negative -> clamp 0
        j2d[1, 2] = torch.tensor([4.0 * h + 9, 2.0])   # beyond -> clamp h-1
        j2d[2, 1] = j2d[2, 0]                          # two joints on one pixel
        vis = (torch.rand(B, J) < 0.8).int()
        vis[3] = 0                                     # image with no visible joint
        vis[0, 0] = 1
        use_depth = torch.tensor([1, 0, 1, 1])
        import torch.nn as nn
        crit = [nn.CrossEntropyLoss(), nn.CrossEntropyLoss()]
        losses, accs = tr._compute_joints_pri3d_loss_accuracy(
            m1, m2, g3, crit, j2d, vis, use_depth=use_depth)
        gm1, gm2, gg3 = torch.autograd.grad(sum(losses), [m1, m2, g3])
        p = 'J%d_' % J
        arrays.update({p + 'map1': m1, p + 'map2': m2, p + 'feat3': g3, p + 'joints2d': j2d,
                       p + 'joints_vis': vis, p + 'use_depth': use_depth,
                       p + 'losses': np.array([f32(l) for l in losses], np.float32),
                       p + 'accs': np.array([f32(a) for a in accs], np.float32),
                       p + 'grad_map1': gm1, p + 'grad_map2': gm2, p + 'grad_feat3': gg3})
        # float64 joints (the loader yields doubles, datasets/dataset.py:594-596)
        losses64, _ = tr._compute_joints_pri3d_loss_accuracy(
            m1, m2, g3, crit, j2d.double(), vis, use_depth=use_depth)
        arrays[p + 'losses_f64joints'] = np.array([f32(l) for l in losses64], np.float32)
    # all-ignored target -> NaN (torch CE semantics)
    torch.manual_seed(9)
    tr = make_trainer()
    import torch.nn as nn
    crit = [nn.CrossEntropyLoss(), nn.CrossEntropyLoss()]
    m = torch.randn(2, 128, 8, 8)
    lnan, _ = tr._compute_joints_pri3d_loss_accuracy(
        m, m, torch.randn(2, 16, 128), crit, torch.rand(2, 16, 2) * 32,
        torch.ones(2, 16).int(), use_depth=torch.zeros(2, dtype=torch.long))
    arrays['allignored_depth_loss_isnan'] = np.array(bool(torch.isnan(lnan[1])))
    npz('joint_nce', temperature=0.07, **arrays)


def gen_scl():
    from learning.segment_trainer import SegTrainer as SegmentTrainer  # sibling copy: use_rgb=None semantics
    arrays = {}
    torch.manual_seed(4242)
    B, C, h, J = 4, 128, 8, 16
    tr = make_trainer()
    st = SegmentTrainer.__new__(SegmentTrainer)
    st.args = tr.args
    m1 = torch.randn(B, C, h, h, requires_grad=True)
    m2 = torch.randn(B, C, h, h, requires_grad=True)
    j2d = torch.rand(B, J, 2) * (4 * h)
    j2d[0, 0] = torch.tensor([-1.0, 40.0])
    vis = (torch.rand(B, J) < 0.8).int()
    use_depth = torch.tensor([1, 0, 1, 1])
    use_rgb = torch.tensor([1, 1, 0, 1])
    # (a) reference contrast_trainer with a use_rgb tensor
    la, _ = tr._compute_cross_subject_joints_pri3d_loss(
        m1, m2, None, None, j2d, vis, use_depth=use_depth, use_rgb=use_rgb)
    ga = torch.autograd.grad(la[0], [m1, m2])
    # (b) use_rgb=None semantics from learning/segment_trainer.py:601-606
    lb, _ = st._compute_cross_subject_joints_pri3d_loss(
        m1, m2, None, None, j2d, vis, use_depth=use_depth, use_rgb=None)
    gb = torch.autograd.grad(lb[0], [m1, m2])
    # (c) early-out
    lc, _ = tr._compute_cross_subject_joints_pri3d_loss(
        m1, m2, None, None, j2d, vis, use_depth=torch.zeros(B, dtype=torch.long), use_rgb=use_rgb)
    arrays.update(map1=m1, map2=m2, joints2d=j2d, joints_vis=vis, use_depth=use_depth, use_rgb=use_rgb,
                  loss_with_rgb=np.float32(f32(la[0])), grad1_with_rgb=ga[0], grad2_with_rgb=ga[1],
                  loss_rgb_none=np.float32(f32(lb[0])), grad1_rgb_none=gb[0], grad2_rgb_none=gb[1],
                  n_early_out=len(lc), early_out=np.array([f32(l) for l in lc], np.float32))
    npz('scl', temperature=0.07, **arrays)


# --------------------------------------------------------------------------- #
# 7. model surface (networks/build_backbone.py) -- keys/shapes + a pinned forward
# --------------------------------------------------------------------------- #
def deterministic_fill(state_dict):
    """Name-keyed deterministic weights, reproducible without the reference."""
    import zlib
    out = {}
    for k, v in state_dict.items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7fffffff)
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
        elif k.endswith('running_var'):
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith('running_mean'):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        elif v.dim() == 1 and k.endswith('.weight'):     # norm scale
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith('.bias'):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        elif k.endswith('.e'):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            fan_in = max(1, int(np.prod(v.shape[1:]))) if v.dim() > 1 else 1
            out[k] = torch.randn(v.shape, generator=g) * (1.0 / np.sqrt(fan_in))
    return out


def gen_model():
    from networks.build_backbone import build_model
    for skel, J in (('mpii', 16), ('coco_reduce', 13)):
        opt = argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                                 in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                                 skeleton_meta_name=skel, IN_Pretrain=None, depth_Pretrain=None, mem='bank')
        model, ema = build_model(opt)
        sd = model.state_dict()
        keys = list(sd.keys())
        shapes = [list(v.shape) for v in sd.values()]
        model.load_state_dict(deterministic_fill(sd))
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 6, 64, 64, generator=g)
        s = torch.rand(2, J, 2, generator=g) * 2 - 1
        arrays = dict(keys=np.array(keys), shapes=np.array([str(s_) for s_ in shapes]),
                      n_params=sum(p.numel() for p in model.parameters()), x=x, s=s)
        for mode in ('eval', 'train'):
            getattr(model, mode)()
            with torch.no_grad():
                f1, f2, f3, f, aux = model(x, s, return_fm=True)
            arrays.update({mode + '_f': f, mode + '_feat3': f3,
                           mode + '_lm1_slice': aux['linear_merge1'][:, :8, ::5, ::5],
                           mode + '_lm2_slice': aux['linear_merge2'][:, :8, ::5, ::5],
                           mode + '_feat1_3': f1[3], mode + '_feat2_0_slice': f2[0][:, :, ::7, ::7]})
        npz('model_hrnet_w18_' + skel, **arrays)


def gen_model_bwd():
    """Backward of the reference model (encoders + SemGCN + heads + 1x1 projections) on CPU, train mode, for the
    ``model`` fixture's inputs and deterministic weights: the loss is <f, cf> + <feat3, c3> + <lm1, c1> + <lm2, c2>
    with seeded cotangents (cf, c3 stored; c1, c2 re-drawn from the same generator by the test).  Stored: d loss / d skeleton (full), every SemGCN / head / projection gradient in
    full, and for EVERY parameter its gradient's L2 norm and its inner product with a name-keyed random vector (the
    same generator as ``deterministic_fill``, seed + 1) -- enough to pin the encoder programs' backward without
    storing 78 MB.  Reference: networks/SGCN/sem_graph_conv.py:34-48, sem_gcn.py:60-95, build_backbone.py:256-303."""
    import zlib
    from networks.build_backbone import build_model
    for skel, J in (('mpii', 16), ('coco_reduce', 13)):
        opt = argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                                 in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                                 skeleton_meta_name=skel, IN_Pretrain=None, depth_Pretrain=None, mem='bank')
        model, _ = build_model(opt)
        model.load_state_dict(deterministic_fill(model.state_dict()))
        model.train()
        # B = 4 at 128 x 128 (not gen_model's 2 x 64 x 64): the coarsest HRNet branch then normalises over 4 x 4 x 4 values
        # instead of 2 x 2 x 2, and a 360-layer reverse chain through train-mode BatchNorm stays well conditioned in fp32.
        # The inputs are seeded, not stored (like the weights): x alone would be 1.5 MB of noise.
        g = torch.Generator().manual_seed(7)
        x = torch.randn(4, 6, 128, 128, generator=g)
        s = (torch.rand(4, J, 2, generator=g) * 2 - 1).requires_grad_(True)
        f1, f2, f3, f, aux = model(x, s, return_fm=True)
        gc = torch.Generator().manual_seed(11)
        cf = torch.randn(f.shape, generator=gc)
        c3 = torch.randn(f3.shape, generator=gc) * 0.1
        c1 = torch.randn(aux['linear_merge1'].shape, generator=gc) * 0.05
        c2 = torch.randn(aux['linear_merge2'].shape, generator=gc) * 0.05
        loss = (f * cf).sum() + (f3 * c3).sum() + (aux['linear_merge1'] * c1).sum() + (aux['linear_merge2'] * c2).sum()
        loss.backward()
        names, norms, dots = [], [], []
        arrays = dict(s=s.detach(), cf=cf, c3=c3, loss=loss.detach(), grad_s=s.grad, x_checksum=x.double().sum())
        for k, p_ in model.named_parameters():
            assert p_.grad is not None, k
            gg = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 1) & 0x7fffffff)
            r = torch.randn(p_.shape, generator=gg)
            names.append(k)
            norms.append(float(p_.grad.double().norm()))
            dots.append(float((p_.grad.double() * r.double()).sum()))
            if not k.startswith(('encoder1.', 'encoder2.')):
                arrays['g:' + k] = p_.grad
        arrays.update(names=np.array(names), norms=np.array(norms), dots=np.array(dots))
        npz('model_bwd_hrnet_w18_' + skel, **arrays)


def gen_model_pn():
    """HRNetPN arch: state_dict keys/shapes (forward / backward: gen_pointnet2_msg, gen_model_pn_fwd, gen_model_pn_bwd)."""
    from networks.build_backbone import build_model
    model, _ = build_model(pn_opt())
    sd = model.state_dict()
    npz('model_hrnetpn_w18_keys', keys=np.array(list(sd.keys())), shapes=np.array([str(list(v.shape)) for v in sd.values()]),
        n_params=sum(p.numel() for p in model.parameters()))


# --------------------------------------------------------------------------- #
# 7b. the PointNet++ MODULE layer and the HRNetPN encoder
#     networks/pointnet2/pointnet2_modules.py:19-55 (SA-MSG forward), pytorch_utils.py:5-33 (SharedMLP),
#     networks/pointnet2_msg.py:79-95, networks/build_backbone.py:379-455 (depth2pts / pts2depth), :457-514 (forward)
#     The reference's Python modules run UNMODIFIED; the nine ``pointnet2_cuda.*_wrapper`` entry points they call are
#     bound to oracle/pointnet2_shim.py -- the C restatement of the reference's kernels (FMA contract = the reference's
#     nvcc -O2 build), itself pinned against those kernels compiled for gfx950 (oracle/_ref, tests/test_pointnet2_ref_gpu.py).
# --------------------------------------------------------------------------- #
def pn_opt(skel='mpii'):
    return argparse.Namespace(modal='RGBD2S', arch='HRNetPN', jigsaw=False, head='linear', feat_dim=128,
                              in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                              skeleton_meta_name=skel, IN_Pretrain=None, depth_Pretrain=None, mem='bank')


def bind_point_ops():
    """``pointnet2_cuda`` := oracle/pointnet2_shim (CPU, FMA contract) inside the reference's pointnet2_utils, and the
    ``torch.cuda.{Int,Float}Tensor`` constructors the reference allocates its outputs with := their CPU twins
    (pointnet2_utils.py:25-26, 55, 67, 94-95, 128, 146, 172, 190, 218)."""
    repo = os.path.dirname(os.path.dirname(OUT))
    if repo not in sys.path:
        sys.path.append(repo)
    from oracle import pointnet2_shim, pointnet2_oracle
    pointnet2_oracle.set_contract('fma')
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    from networks.pointnet2 import pointnet2_utils as ref_utils
    ref_utils.pointnet2 = pointnet2_shim
    return ref_utils


def surface_cloud(B, N, seed):
    """A depth-camera-like cloud in metres: a smooth surface over [-0.5, 0.5] x [-1, 1] with 1 cm noise, one third of
    the points exact duplicates (depth2pts samples pixels WITH replacement, build_backbone.py:427) -- level-1 balls of
    2.5 cm hold a handful of points (padded groups), 12.5 cm balls overflow their 32 slots."""
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, N, 2, generator=g) * torch.tensor([1.0, 2.0]) - torch.tensor([0.5, 1.0])
    z = 0.2 * torch.sin(3 * xy[..., 0]) * torch.cos(2 * xy[..., 1]) + 0.01 * torch.randn(B, N, generator=g)
    pts = torch.cat([xy, z.unsqueeze(-1)], -1)
    src = torch.randint(0, N, (B, N // 3), generator=g)
    for b in range(B):
        pts[b, :N // 3] = pts[b, src[b]]
    return pts.contiguous()


def grad_summary(named_params, full_below=0, full_prefixes=()):
    """names / L2 norms / projections on name-keyed random vectors (seed crc32(name) + 1, as gen_model_bwd), and the
    gradient itself for small tensors or the given prefixes."""
    import zlib
    names, norms, dots, full = [], [], [], {}
    for k, p_ in named_params:
        assert p_.grad is not None, k
        gg = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 1) & 0x7fffffff)
        r = torch.randn(p_.shape, generator=gg)
        names.append(k)
        norms.append(float(p_.grad.double().norm()))
        dots.append(float((p_.grad.double() * r.double()).sum()))
        if p_.numel() <= full_below or k.startswith(tuple(full_prefixes)):
            full['g:' + k] = p_.grad
    return dict(names=np.array(names), norms=np.array(norms), dots=np.array(dots), **full)


def gen_pointnet2_msg():
    """The reference's ``Pointnet2MSG(input_channels=0)`` (4 SA-MSG levels, 4 FP levels) on a 2 x 4096-point cloud with
    name-keyed deterministic weights: forward in eval and train mode (per-level centres and features, the output), and
    the reference's backward in train mode for a seeded cotangent (every parameter gradient as norm + projection, small
    ones in full)."""
    bind_point_ops()
    from networks.pointnet2_msg import Pointnet2MSG
    net = Pointnet2MSG(input_channels=0)
    net.load_state_dict(deterministic_fill(net.state_dict()))
    cloud = surface_cloud(2, 4096, 21)
    arrays = dict(cloud=cloud, keys=np.array(list(net.state_dict().keys())))
    levels = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: levels.append(out)) for m in net.SA_modules]
    for mode in ('eval', 'train'):
        getattr(net, mode)()
        del levels[:]
        with torch.no_grad():
            out = net(cloud)
        gp = torch.Generator().manual_seed(31)
        proj = torch.randn(out.shape, generator=gp)
        arrays.update({mode + '_out_slice': out[:, ::2, ::32], mode + '_out_norm': out.double().norm(),
                       mode + '_out_dot': (out.double() * proj.double()).sum()})
        for k, (xyz_k, feat_k) in enumerate(levels):
            arrays['%s_l%d_xyz' % (mode, k + 1)] = xyz_k if k else xyz_k[:, ::16]      # FPS picks, chained through the levels
            arrays['%s_l%d_feat_slice' % (mode, k + 1)] = feat_k[:, ::4, ::max(1, feat_k.shape[2] // 64)]
            arrays['%s_l%d_feat_norm' % (mode, k + 1)] = feat_k.double().norm()
    for h_ in hooks:
        h_.remove()
    # backward, train mode (running statistics of the eval/train passes above are irrelevant to batch-stat BN)
    net.load_state_dict(deterministic_fill(net.state_dict()))
    net.train()
    out = net(cloud)
    gc = torch.Generator().manual_seed(33)
    cot = torch.randn(out.shape, generator=gc) * 0.1
    loss = (out * cot).sum()
    loss.backward()
    arrays.update(loss=loss.detach(), **grad_summary(net.named_parameters(), full_below=2048))
    sd = net.state_dict()
    arrays['bn_running_mean_after'] = sd['SA_modules.0.mlps.0.layer0.bn.bn.running_mean']
    arrays['bn_running_var_after'] = sd['SA_modules.0.mlps.0.layer0.bn.bn.running_var']
    npz('pointnet2_msg', **arrays)


def pn_inputs(*a, **k):
    """Seeded HRNetPN inputs: tests/golden/pn_inputs.py (shared with the tests, which re-create them instead of loading
    1.5 MB of noise)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('pn_inputs', os.path.join(OUT, 'pn_inputs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.pn_inputs(*a, **k)


class capture_multinomial(object):
    """Record what ``Tensor.multinomial`` returns inside the reference's depth2pts (build_backbone.py:427)."""

    def __enter__(self):
        self.drawn = []
        self.orig = torch.Tensor.multinomial
        cap = self

        def wrapped(t, *a, **k):
            r = cap.orig(t, *a, **k)
            cap.drawn.append(r.clone())
            return r
        torch.Tensor.multinomial = wrapped
        return self

    def __exit__(self, *exc):
        torch.Tensor.multinomial = self.orig


def replay_multinomial(ind):
    """Make the next ``Tensor.multinomial`` calls return the recorded draw (the second forward sees the same cloud)."""
    class _R(object):
        def __enter__(self_):
            self_.orig = torch.Tensor.multinomial
            torch.Tensor.multinomial = lambda t, *a, **k: ind.clone()

        def __exit__(self_, *exc):
            torch.Tensor.multinomial = self_.orig
    return _R()


def gen_model_pn_fwd():
    """The reference's ``CMC3HRNetSGCNPN2SingleHead`` (HRNet-w18 + Pointnet2MSG + SemGCN, build_backbone.py:305-514)
    forward with ``return_fm=True`` in eval and train mode: B = 3 at 64 x 64, the third sample without depth (its cloud is
    all zeros, :399-443).  The pixel draw of depth2pts is recorded (``ind``) and replayed for the second mode."""
    bind_point_ops()
    from networks.build_backbone import build_model
    model, _ = build_model(pn_opt())
    model.load_state_dict(deterministic_fill(model.state_dict()))
    x, s, mask, grid_xy, oh, ow, mean = pn_inputs(3, 64, 16, 41, empty=(2,))
    arrays = dict(x=x, s=s, depth_mask=mask, grid_xy=grid_xy, original_h=oh, original_w=ow, mean=mean)
    ind = None
    for mode in ('eval', 'train'):
        getattr(model, mode)()
        torch.manual_seed(43)
        with torch.no_grad():
            if ind is None:
                with capture_multinomial() as cap:
                    f1, f2, f3, f, aux = model(x, s, mask, grid_xy, oh, ow, mean, return_fm=True)
                assert len(cap.drawn) == 1
                ind = cap.drawn[0]
                x1, x2 = torch.split(x, [3, 3], dim=1)
                with replay_multinomial(ind):
                    sample, full, _ = model.depth2pts(x2, mask, grid_xy, oh, ow, mean)
                arrays.update(ind=ind.int(), cloud_sample_slice=sample[:, :, ::8], cloud_sample_sum=sample.double().sum(),
                              cloud_full_slice=full[:, :, ::8], cloud_full_sum=full.double().sum())
            else:
                with replay_multinomial(ind):
                    f1, f2, f3, f, aux = model(x, s, mask, grid_xy, oh, ow, mean, return_fm=True)
        assert f2.shape == (3, 128, 4096) and aux['linear_merge2'].shape == (3, 128, 16, 16), (f2.shape, aux['linear_merge2'].shape)
        arrays.update({mode + '_f': f, mode + '_feat3': f3, mode + '_feat2_slice': f2[:, ::2, ::32],
                       mode + '_feat2_norm': f2.double().norm(),
                       mode + '_lm1_slice': aux['linear_merge1'][:, :8, ::5, ::5],
                       mode + '_lm2': aux['linear_merge2'][:, ::2],
                       mode + '_feat1_3': f1[3]})
    npz('model_hrnetpn_w18_mpii', **arrays)


def gen_model_pn_bwd():
    """The reference HRNetPN model's backward on CPU, train mode: B = 4 at 128 x 128 (gen_model_bwd's reason: the
    coarsest HRNet branch must normalise over more than 8 values), sample 1 without depth.  Loss = <f, cf> + <feat3, c3>
    + <lm1, c1> + <lm2, c2> with seeded cotangents.  Stored: the inputs that are not re-creatable from a seed alone
    (the pixel draw), d loss / d skeleton, norm + projection of every parameter gradient, and the gradients of the
    PointNet++ encoder's small tensors, encoder2_linear, the heads and the SemGCN in full."""
    bind_point_ops()
    from networks.build_backbone import build_model
    model, _ = build_model(pn_opt())
    model.load_state_dict(deterministic_fill(model.state_dict()))
    model.train()
    x, s, mask, grid_xy, oh, ow, mean = pn_inputs(4, 128, 16, 47, empty=(1,))
    s.requires_grad_(True)
    torch.manual_seed(49)
    with capture_multinomial() as cap:
        f1, f2, f3, f, aux = model(x, s, mask, grid_xy, oh, ow, mean, return_fm=True)
    assert len(cap.drawn) == 1
    gc = torch.Generator().manual_seed(53)
    cf = torch.randn(f.shape, generator=gc)
    c3 = torch.randn(f3.shape, generator=gc) * 0.1
    c1 = torch.randn(aux['linear_merge1'].shape, generator=gc) * 0.05
    c2 = torch.randn(aux['linear_merge2'].shape, generator=gc) * 0.05
    terms = [f * cf, f3 * c3, aux['linear_merge1'] * c1, aux['linear_merge2'] * c2]
    loss = terms[0].sum() + terms[1].sum() + terms[2].sum() + terms[3].sum()
    loss.backward()
    arrays = dict(ind=cap.drawn[0].int(), s=s.detach(), cf=cf, c3=c3, loss=loss.detach(), grad_s=s.grad,
                  mass=sum(t.detach().abs().sum() for t in terms), x_checksum=x.double().sum(), mean=mean,
                  lm2_slice=aux['linear_merge2'].detach()[:, ::4, ::3, ::3], f=f.detach())
    full = tuple(k for k, p_ in model.named_parameters()
                 if not k.startswith('encoder1.') and (not k.startswith('encoder2.') or p_.numel() <= 2048))
    arrays.update(grad_summary(model.named_parameters(), full_prefixes=full))
    npz('model_bwd_hrnetpn_w18_mpii', **arrays)


# --------------------------------------------------------------------------- #
# 8. options surface (options/train_options.py)
# --------------------------------------------------------------------------- #
def gen_options():
    from options.train_options import TrainOptions
    import tempfile
    import json
    tmp = tempfile.mkdtemp()
    cases = {
        'stage1': ['--method', 'CMCRGBD2S', '--arch', 'HRNet', '--width', '18', '--in_channel_list', '3,3',
                   '--batch_size', '8', '--nce_k', '256', '--cosine', '--lr_decay_epochs', '30,60'],
        'custom_warm': ['--method', 'Customize', '--batch_size', '512', '--epochs', '600', '--cosine',
                        '--learning_rate', '0.06', '--modal', 'CMC', '--mem', 'moco'],
        'infomin': ['--method', 'InfoMin', '--amp', '--opt_level', 'O1', '--tag', 'x'],
    }
    out = {}
    for name, argv in cases.items():
        sys.argv = ['main_contrast.py', '--model_path', tmp, '--tb_path', tmp] + argv
        opt = TrainOptions().parse()
        d = {k: v for k, v in vars(opt).items() if k not in ('model_path', 'tb_path', 'model_folder', 'tb_folder')}
        out[name] = dict(argv=argv, opt=d)
    with open(os.path.join(OUT, 'options_cases.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote options_cases.json')


# --------------------------------------------------------------------------- #
# 9. 2-step traces of the reference's OWN training loops (SURVEY 8c "harness rows")
#    learning/contrast_trainer.py:532-640 (_train_mem_skeleton3d, stage 1, use_rgb + use_depth)
#    learning/contrast_trainer.py:894-1039 (_train_bank_joints_pri3d_cmc3, stage 2)
#    driven with the reference CMCMem3, torch.optim.SGD and a stand-in encoder (tests/golden/standin.py).
# --------------------------------------------------------------------------- #
def gen_trace():
    import importlib.util
    import torch.distributed as dist
    import torch.nn as nn
    from memory.mem_bank import CMCMem3
    from learning.contrast_trainer import ContrastTrainer
    spec = importlib.util.spec_from_file_location('standin', os.path.join(OUT, 'standin.py'))
    standin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(standin)
    if not dist.is_initialized():                  # the loops call dist.all_gather (world size 1 here)
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29871', rank=0, world_size=1)

    B, n, K, H, J, S, steps = 6, 48, 20, 32, 16, 12, 2
    T, mom, lr = 0.07, 0.5, 0.05
    for stage in (1, 2):
        torch.manual_seed(900 + stage)
        model = standin.StandInEncoder()
        mem = CMCMem3(128, n, K, T, mom)
        opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
        batches = standin.make_batches(steps, B, n, H, J, seed=40 + stage)
        tr = ContrastTrainer.__new__(ContrastTrainer)
        tr.args = argparse.Namespace(gpu=None, arch='HRNet', modality_missing=1, amp=False, local_rank=0,
                                     print_freq=10 ** 6, warm=False, jigsaw=False, temperature=T,
                                     pri3d_num_samples_per_image=S)
        arrays = dict(B=B, n=n, K=K, H=H, J=J, S=S, T=T, m=mom, lr=lr, momentum=0.9, weight_decay=1e-4, steps=steps,
                      bank0_1=mem.memory_1.clone(), bank0_2=mem.memory_2.clone(), bank0_3=mem.memory_3.clone())
        for k, v in model.state_dict().items():
            arrays['w0_' + k] = v.clone()
        for t, b in enumerate(batches):
            for i, item in enumerate(b):
                arrays['s%d_data%d' % (t, i)] = item
        rec = {'idx': [], 'ind': [], 'bank': [], 'dense': [], 'joint': [], 'scl': [], 'after': []}

        orig_draw = mem.multinomial.draw
        mem.multinomial.draw = lambda N: rec['idx'].append(orig_draw(N).clone()) or rec['idx'][-1].clone()
        orig_mn = torch.Tensor.multinomial

        def mn(self, *a, **k):
            out = orig_mn(self, *a, **k)
            rec['ind'].append(out.clone())
            return out

        def wrap(name, key):
            fn = getattr(tr, name)

            def inner(*a, **k):
                out = fn(*a, **k)
                rec[key].append([[f32(v) for v in part] for part in out])
                return out
            setattr(tr, name, inner)
        wrap('_compute_loss_accuracy', 'bank')
        wrap('_compute_soft_pri3d_loss_accuracy', 'dense')
        wrap('_compute_joints_pri3d_loss_accuracy', 'joint')
        wrap('_compute_cross_subject_joints_pri3d_loss', 'scl')
        orig_step = opt.step

        def step(*a, **k):
            out = orig_step(*a, **k)
            rec['after'].append(([mem.memory_1.clone(), mem.memory_2.clone(), mem.memory_3.clone()],
                                 {k2: v.clone() for k2, v in model.state_dict().items()}))
            return out
        opt.step = step
        torch.Tensor.multinomial = mn
        try:
            if stage == 1:
                outs = tr._train_mem_skeleton3d(1, batches, model, mem, nn.CrossEntropyLoss(), opt)
            else:
                outs = tr._train_bank_joints_pri3d_cmc3(1, batches, model, mem, nn.CrossEntropyLoss(),
                                                        [nn.CrossEntropyLoss(), nn.CrossEntropyLoss()], opt)
        finally:
            torch.Tensor.multinomial = orig_mn
        assert len(rec['after']) == steps and len(rec['idx']) == steps
        arrays['epoch_outs'] = np.array([float(o) for o in outs], np.float64)
        for t in range(steps):
            arrays['s%d_idx' % t] = rec['idx'][t].view(B, K + 1)
            arrays['s%d_bank_losses' % t] = np.array(rec['bank'][t][0], np.float32)
            arrays['s%d_bank_accs' % t] = np.array(rec['bank'][t][1], np.float32)
            if stage == 2:
                arrays['s%d_sample_ind' % t] = rec['ind'][t]
                arrays['s%d_dense' % t] = np.array(rec['dense'][t][0] + rec['dense'][t][1], np.float32)
                arrays['s%d_joint' % t] = np.array(rec['joint'][t][0] + rec['joint'][t][1], np.float32)
                arrays['s%d_scl' % t] = np.array(rec['scl'][t][0], np.float32)
            banks, sd = rec['after'][t]
            for i, bk in enumerate(banks):
                arrays['s%d_bank_%d' % (t, i + 1)] = bk
                arrays['s%d_bank_%d_checksum' % (t, i + 1)] = np.float64(bk.double().sum().item())
            for k, v in sd.items():
                arrays['s%d_w_%s' % (t, k)] = v
        npz('trace_stage%d' % stage, **arrays)


# --------------------------------------------------------------------------- #
# 9b. 3-step trace of the reference's MoCo loop (learning/contrast_trainer.py:255-389 _train_moco, :167-210
#     _shuffle_bn, :1041-1045 momentum_update) with the reference CMCMoCo (memory/mem_moco.py:91-142), torch SGD and
#     the stand-in CMC encoder pair of tests/golden/standin.py.  World size 1: the shuffle is a permutation of the
#     local batch.  K = 20 with B = 6: the ring pointer wraps inside the trace.
# --------------------------------------------------------------------------- #
def gen_trace_moco():
    import importlib.util
    import torch.distributed as dist
    import torch.nn as nn
    from memory.mem_moco import CMCMoCo
    from learning.contrast_trainer import ContrastTrainer
    spec = importlib.util.spec_from_file_location('standin', os.path.join(OUT, 'standin.py'))
    standin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(standin)
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29872', rank=0, world_size=1)
    B, K, H, D, steps = 6, 20, 16, 64, 4   # D: the queue kernels take 64 or 128
    T, alpha, lr = 0.2, 0.9, 0.05
    torch.manual_seed(931)
    model = standin.StandInMoCoEncoder(D=D)
    model_ema = standin.StandInMoCoEncoder(D=D)
    mem = CMCMoCo(D, K, T)
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    batches = standin.make_moco_batches(steps, B, H, seed=77)
    tr = ContrastTrainer.__new__(ContrastTrainer)
    tr.args = argparse.Namespace(gpu=None, amp=False, local_rank=0, node_rank=0, ngpus_per_node=1, print_freq=10 ** 6,
                                 warm=False, jigsaw=False, modal='CMC', alpha=alpha, beta=0.5)
    tr.local_group = dist.new_group([0])
    ContrastTrainer.momentum_update(model, model_ema, 0)            # main_contrast.py / wrap_up: ema <- model

    class Wrapped(nn.Module):                                       # the loop updates `model.module` (DDP)
        def __init__(self, m):
            super().__init__()
            self.module = m

        def forward(self, *a, **k):
            return self.module(*a, **k)
    wrapped = Wrapped(model)
    arrays = dict(B=B, K=K, H=H, D=D, T=T, alpha=alpha, lr=lr, momentum=0.9, weight_decay=1e-4, steps=steps,
                  queue0_1=mem.memory_1.clone(), queue0_2=mem.memory_2.clone())
    for k, v in model.state_dict().items():
        arrays['w0_' + k] = v.clone()
    for k, v in model_ema.state_dict().items():
        arrays['e0_' + k] = v.clone()
    for t, b in enumerate(batches):
        arrays['s%d_data0' % t], arrays['s%d_data1' % t] = b[0], b[1]
    rec = {'perm': [], 'ce': [], 'logits': [], 'after': []}
    orig_perm = torch.randperm

    def perm(*a, **k):
        out = orig_perm(*a, **k)
        rec['perm'].append(out.clone())
        return out
    fn = tr._compute_loss_accuracy

    def ce(*a, **k):
        out = fn(*a, **k)
        rec['logits'].append([l.detach().clone() for l in k['logits']])
        rec['ce'].append(([f32(v) for v in out[0]], [f32(v) for v in out[1]]))
        return out
    tr._compute_loss_accuracy = ce
    orig_mu = ContrastTrainer.momentum_update

    def mu(m, e, a):                                                # called last in an iteration (:372)
        orig_mu(m, e, a)
        rec['after'].append(({k: v.clone() for k, v in m.state_dict().items()}, {k: v.clone() for k, v in e.state_dict().items()},
                             mem.memory_1.clone(), mem.memory_2.clone(), mem.index))
    tr.momentum_update = mu
    torch.randperm = perm
    try:
        outs = tr._train_moco(1, batches, wrapped, model_ema, mem, nn.CrossEntropyLoss(), opt)
    finally:
        torch.randperm = orig_perm
    assert len(rec['after']) == steps and len(rec['perm']) == steps
    arrays['epoch_outs'] = np.array([float(o) for o in outs], np.float64)
    for t in range(steps):
        arrays['s%d_shuffle_ids' % t] = rec['perm'][t]
        arrays['s%d_logits1' % t], arrays['s%d_logits2' % t] = rec['logits'][t][0], rec['logits'][t][1]
        arrays['s%d_losses' % t] = np.array(rec['ce'][t][0], np.float32)
        arrays['s%d_accs' % t] = np.array(rec['ce'][t][1], np.float32)
        sd, se, q1, q2, index = rec['after'][t]
        arrays['s%d_queue_1' % t], arrays['s%d_queue_2' % t], arrays['s%d_index' % t] = q1, q2, index
        for k, v in sd.items():
            arrays['s%d_w_%s' % (t, k)] = v
        for k, v in se.items():
            arrays['s%d_e_%s' % (t, k)] = v
    npz('trace_moco', **arrays)


# --------------------------------------------------------------------------- #
# 10. dataset tuple producers (datasets/dataset.py:306-617, datasets/mpii_utils.py:14-65): the numpy / torch
#     arithmetic AROUND the image decoding.  cv2 / torchvision / json_tricks / pycocotools are not in the image:
#     process-local stub modules let `datasets.dataset` import; the functions driven here never touch them, except
#     get_affine_transform (calls cv2.getAffineTransform = the exact solve of three point pairs, stubbed with
#     numpy.linalg.solve) and the *_getitem__ branches whose decoded image / super().__getitem__ result is injected.
# --------------------------------------------------------------------------- #
def install_dataset_stubs():
    import json
    np.float = float                                   # numpy 2 dropped the alias the reference uses (:341-353)
    cv2 = types.ModuleType('cv2')

    def get_affine(src, dst):
        a = np.zeros((6, 6), np.float64)
        b = np.zeros(6, np.float64)
        for i in range(3):
            a[2 * i] = [src[i][0], src[i][1], 1, 0, 0, 0]
            a[2 * i + 1] = [0, 0, 0, src[i][0], src[i][1], 1]
            b[2 * i], b[2 * i + 1] = dst[i][0], dst[i][1]
        return np.linalg.solve(a, b).reshape(2, 3)
    cv2.getAffineTransform = get_affine
    cv2.INTER_LINEAR, cv2.IMREAD_COLOR, cv2.IMREAD_IGNORE_ORIENTATION, cv2.COLOR_BGR2RGB = 1, 1, 128, 4
    sys.modules['cv2'] = cv2
    tv = types.ModuleType('torchvision')
    tvd = types.ModuleType('torchvision.datasets')
    tvd.ImageFolder = type('ImageFolder', (), {})
    tvt = types.ModuleType('torchvision.transforms')
    tvf = types.ModuleType('torchvision.transforms.functional')
    tv.datasets, tv.transforms, tvt.functional = tvd, tvt, tvf
    sys.modules.update({'torchvision': tv, 'torchvision.datasets': tvd, 'torchvision.transforms': tvt,
                        'torchvision.transforms.functional': tvf})
    jt = types.ModuleType('json_tricks')
    jt.load, jt.loads, jt.dump, jt.dumps = json.load, json.loads, json.dump, json.dumps
    sys.modules['json_tricks'] = jt
    pc = types.ModuleType('pycocotools')
    pcc = types.ModuleType('pycocotools.coco')
    pcc.COCO = type('COCO', (), {})
    pc.coco = pcc
    sys.modules.update({'pycocotools': pc, 'pycocotools.coco': pcc})
    for name in ('skimage', 'skimage.io', 'skimage.transform', 'h5py', 'scipy.io'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def gen_dataset():
    import json
    import tempfile
    install_dataset_stubs()
    # the reference's datasets/ has no __init__.py and a `datasets` distribution is installed in this image:
    # load the directory as an explicitly-pathed package so that its relative imports resolve
    import importlib
    pkg = types.ModuleType('refdatasets')
    pkg.__path__ = [os.path.join(REF, 'datasets')]
    sys.modules['refdatasets'] = pkg
    D = importlib.import_module('refdatasets.dataset')
    MU = importlib.import_module('refdatasets.mpii_utils')
    g = np.random.RandomState(77)
    arrays = {}
    # --- MPII annotation records (:330-381)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'annot'))
    anno = []
    for k in range(4):
        joints = (g.rand(16, 2) * 300 + 20).round(2)
        vis = (g.rand(16) < 0.8).astype(int)
        anno.append({'image': 'im%d.jpg' % k, 'center': [float(150 + 10 * k), float(120 - 5 * k)] if k != 2 else [-1.0, -1.0],
                     'scale': float(1.1 + 0.3 * k), 'joints': joints.tolist(), 'joints_vis': vis.tolist()})
    with open(os.path.join(tmp, 'annot', 'train.json'), 'w') as f:
        json.dump(anno, f)
    obj = D.NTUMPIIRGBD3D2DSkeletonGCN.__new__(D.NTUMPIIRGBD3D2DSkeletonGCN)
    db = obj._get_db(tmp, 'train', 16, 'jpg')
    arrays['mpii_anno_json'] = np.array(json.dumps(anno))
    arrays['mpii_center'] = np.stack([r['center'] for r in db])
    arrays['mpii_scale'] = np.stack([r['scale'] for r in db])
    arrays['mpii_joints'] = np.stack([r['joints_3d'] for r in db])
    arrays['mpii_joints_vis'] = np.stack([r['joints_3d_vis'] for r in db])
    arrays['mpii_image'] = np.array([os.path.relpath(r['image'], tmp) for r in db])
    # --- joint bookkeeping
    k25 = g.rand(25, 2).astype(np.float32) * 400
    arrays['kinect25'] = k25
    arrays['kinect2mpii'] = obj.Kinect2MPII(k25)
    obj.flip_pairs = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]
    j16 = g.rand(16, 2).astype(np.float32) * 200
    arrays['joints16'] = j16
    arrays['norm_myway'] = obj.normalize_joints_myway(j16)
    arrays['norm_flipped'] = obj.flip_normalized_joints(obj.normalize_joints_myway(j16).copy())
    vis = g.rand(16) < 0.7
    arrays['vis16'] = vis
    arrays['scale_mpii'] = np.float64(D.generate_scale_mpii(j16.astype(np.float64), vis))
    arrays['scale_mpii_none'] = np.float64(D.generate_scale_mpii(j16.astype(np.float64), np.zeros(16, bool)))
    # --- affine helpers (mpii_utils.py:14-65)
    for k, (c, s, r) in enumerate([((150., 120.), (1.2, 1.2), 0.0), ((88.5, 240.25), (2.0, 2.0), 17.5),
                                   ((10., 10.), (0.7, 0.7), -41.0)]):
        t = MU.get_affine_transform(np.array(c), np.array(s), r, (256, 256))
        arrays['affine%d_in' % k] = np.array([c[0], c[1], s[0], s[1], r])
        arrays['affine%d' % k] = t
        arrays['affine%d_pt' % k] = MU.affine_transform(np.array([31.0, 77.0]), t)
    # --- NTU branch of the GCN tuple (:570-617) with the decoded frame injected
    size = 64
    obj.size, obj.random_flip, obj.random_resized_crop = (size, size), True, True
    obj.db, obj.mpii_num_joints, obj.num_joints = [], 16, 25
    depth = torch.from_numpy((g.rand(size, size) * 3000).astype(np.float32) / 1000.0)
    depth[:, :20] = 0
    rgbd = torch.cat([torch.from_numpy(g.randn(3, size, size).astype(np.float32)), torch.stack([depth] * 3)], 0)
    dloc = (g.rand(25, 2) * np.array([400, 300]) + np.array([500, 300])).astype(np.float32)
    joints3d = torch.from_numpy(g.randn(25, 3).astype(np.float32))
    resize_param = (350, 520, 380, 380, True, 1080, 1920)
    skel = {'joints': [{'d_loc': [list(map(float, p)) for p in dloc]}]}
    orig = D.NTURGBD3DSkeleton.__getitem__
    D.NTURGBD3DSkeleton.__getitem__ = lambda self, index, return_resize_param=False: (rgbd.clone(), index, joints3d, resize_param, skel)
    try:
        out = obj[5]
    finally:
        D.NTURGBD3DSkeleton.__getitem__ = orig
    arrays.update(ntu_rgbd_in=rgbd, ntu_dloc=dloc, ntu_joints3d=joints3d, ntu_resize_param=np.array(resize_param[:4] + resize_param[5:]),
                  ntu_need_flip=resize_param[4], ntu_size=size)
    names = ['rgbd', 'index', 'norm_joints', 'joints3d', 'original_joints2d', 'joints_vis', 'true_depth', 'depth_mask', 'scale']
    for n, v in zip(names, out):
        arrays['ntu_out_' + n] = v if isinstance(v, torch.Tensor) else np.asarray(v)
    # the same frame when the image was NOT mirrored (need_flip False) under --random_flip: the reference tests
    # resize_param[-1] (= original_w, always truthy) at dataset.py:589, so the normalised skeleton is mirrored anyway
    resize_noflip = resize_param[:4] + (False,) + resize_param[5:]
    D.NTURGBD3DSkeleton.__getitem__ = lambda self, index, return_resize_param=False: (rgbd.clone(), index, joints3d, resize_noflip, skel)
    try:
        out_nf = obj[5]
    finally:
        D.NTURGBD3DSkeleton.__getitem__ = orig
    arrays['ntu_noflip_out_norm_joints'] = out_nf[2]
    arrays['ntu_noflip_out_original_joints2d'] = out_nf[4]

    # --- NTU + COCO variant (:622-955): annotation records through the reference's own loader (the pycocotools API
    #     it calls is replaced by a minimal reader of the same json), box -> centre/scale, the two joint reductions,
    #     and the NTU branch of its __getitem__ with the decoded frame injected
    coco_json = {'categories': [{'id': 1, 'name': 'person'}, {'id': 7, 'name': 'other'}], 'images': [], 'annotations': []}
    aid = 0
    for im_id, (w_, h_) in ((9, (640, 480)), (3, (500, 375)), (12, (320, 240))):
        coco_json['images'].append({'id': im_id, 'width': w_, 'height': h_})
        for k in range(3):
            kp = []
            for q in range(17):
                kp += [float(g.randint(0, w_)), float(g.randint(0, h_)), int(g.randint(0, 3))]
            if im_id == 3 and k == 1:
                kp = [0] * 51                                  # person without keypoints: dropped
            box = [float(g.randint(-20, w_ - 50)), float(g.randint(-10, h_ - 50)), float(g.randint(30, 300)), float(g.randint(30, 300))]
            coco_json['annotations'].append({'id': aid, 'image_id': im_id, 'category_id': 7 if (im_id == 12 and k == 0) else 1,
                                             'iscrowd': int(im_id == 9 and k == 2), 'area': 0.0 if (im_id == 12 and k == 2) else 900.0,
                                             'bbox': box, 'keypoints': kp})
            aid += 1
    os.makedirs(os.path.join(tmp, 'annotations'))
    with open(os.path.join(tmp, 'annotations', 'person_keypoints_train2014.json'), 'w') as f:
        json.dump(coco_json, f)

    class MiniCOCO(object):                                     # the five pycocotools calls of dataset.py:631-722
        def __init__(self, path):
            self.d = json.load(open(path))

        def getCatIds(self):
            return [c['id'] for c in self.d['categories']]

        def loadCats(self, ids):
            return [c for c in self.d['categories'] if c['id'] in ids]

        def getImgIds(self):
            return [i['id'] for i in self.d['images']]

        def loadImgs(self, i):
            return [im for im in self.d['images'] if im['id'] == i]

        def getAnnIds(self, imgIds, iscrowd=None):
            return [a['id'] for a in self.d['annotations'] if a['image_id'] == imgIds and (iscrowd is None or bool(a['iscrowd']) == iscrowd)]

        def loadAnns(self, ids):
            return [a for a in self.d['annotations'] if a['id'] in ids]
    D.COCO = MiniCOCO
    cobj = D.NTUCOCORGBD3D2DSkeletonGCN.__new__(D.NTUCOCORGBD3D2DSkeletonGCN)
    cobj.coco_root, cobj.coco_image_set = tmp, 'train2014'
    cobj.coco = MiniCOCO(cobj._get_ann_file_keypoint())
    cats = [c['name'] for c in cobj.coco.loadCats(cobj.coco.getCatIds())]
    cobj.classes = ['__background__'] + cats
    cobj._class_to_ind = dict(zip(cobj.classes, range(len(cobj.classes))))
    cobj._class_to_coco_ind = dict(zip(cats, cobj.coco.getCatIds()))
    cobj._coco_ind_to_class_ind = dict([(cobj._class_to_coco_ind[c], cobj._class_to_ind[c]) for c in cobj.classes[1:]])
    cobj.image_set_index = cobj._load_image_set_index()
    cobj.coco_num_joints, cobj.is_train, cobj.aspect_ratio, cobj.pixel_std, cobj.data_format = 17, True, 1.0, 200, 'jpg'
    cdb = cobj._get_db()
    arrays['coco_anno_json'] = np.array(json.dumps(coco_json))
    arrays['coco_center'] = np.stack([r['center'] for r in cdb])
    arrays['coco_scale'] = np.stack([r['scale'] for r in cdb])
    arrays['coco_joints'] = np.stack([r['joints_3d'] for r in cdb])
    arrays['coco_joints_vis'] = np.stack([r['joints_3d_vis'] for r in cdb])
    arrays['coco_image'] = np.array([os.path.relpath(r['image'], tmp) for r in cdb])
    n17, o17, v17 = g.rand(17, 2), g.rand(17, 2) * 100, g.rand(17) < 0.6
    r = cobj.COCOReduce(n17, o17, v17)
    arrays.update(coco_in_norm=n17, coco_in_orig=o17, coco_in_vis=v17, coco_red_norm=r[0], coco_red_orig=r[1], coco_red_vis=r[2],
                  kinect_reduce=cobj.KinectReduce(k25))
    cobj.size, cobj.random_flip, cobj.random_resized_crop, cobj.db, cobj.num_joints = (size, size), False, True, [], 25
    cobj.flip_pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    D.NTURGBD3DSkeleton.__getitem__ = lambda self, index, return_resize_param=False: (rgbd.clone(), index, joints3d, resize_param, skel)
    try:
        cout = cobj[2]
    finally:
        D.NTURGBD3DSkeleton.__getitem__ = orig
    for n, v in zip(names, cout):
        arrays['ntucoco_out_' + n] = v if isinstance(v, torch.Tensor) else np.asarray(v)

    # --- NTU-segmentation variant (:957-1120): label mapping, modality masking, grid_xy, mean -- the tail of its
    #     __getitem__ with the decoded frame injected; TF.resized_crop on PIL images = crop + resize (stubbed so)
    from PIL import Image as PILImage
    D.TF.resized_crop = lambda img, i, j, h, w, size, interpolation=PILImage.BILINEAR: img.crop((j, i, j + w, i + h)).resize(
        tuple(size[::-1]), interpolation)
    lab = np.kron(g.choice(np.array([0, 1, 2, 3, 6, 7, 8, 17, 18, 19, 25, 26, 27, 32, 33, 34, 38, 39, 43, 44, 46, 49, 50, 56, 58]),
                           size=(36, 64)), np.ones((30, 30), np.int64)).astype(np.uint8)          # 1080 x 1920 in 30-pixel blocks
    lab_path = os.path.join(tmp, 'label.png')
    PILImage.fromarray(lab).save(lab_path)
    sobj = D.NTURGBDSegJoint.__new__(D.NTURGBDSegJoint)
    sobj.size, sobj.random_flip, sobj.random_resized_crop, sobj.mpii_num_joints = (size, size), False, True, 16
    sobj.only_seg, sobj.split, sobj.seg_gt_list = False, 3, [lab_path, lab_path]
    sobj.label_mapper = np.arange(60)
    for i_, l_ in enumerate([0, 1, 2, 3, 6, 7, 8, 17, 18, 19, 25, 26, 27, 32, 33, 34, 38, 39, 43, 44, 46, 49, 50, 56, 58]):
        sobj.label_mapper[l_] = i_
    D.NTURGBD3DSkeleton.__getitem__ = lambda self, index, return_resize_param=False: (rgbd.clone(), index, joints3d, resize_param, skel)
    seg_names = names + ['label', 'true_label', 'true_rgb', 'grid_xy', 'original_h', 'original_w', 'mean']
    try:
        for tag, idx, md, mr in (('plain', 1, False, False), ('parsing', 4, False, False), ('nodepth', 4, True, False),
                                 ('norgb', 3, False, True)):
            sobj.mask_seg_depth, sobj.mask_seg_rgb = md, mr
            sout = sobj[idx]
            for n, v in zip(seg_names, sout):
                arrays['seg_%s_%s' % (tag, n)] = v if isinstance(v, torch.Tensor) else np.asarray(v)
    finally:
        D.NTURGBD3DSkeleton.__getitem__ = orig
    arrays['seg_label_png'] = lab
    npz('dataset_tuple', **arrays)


if __name__ == '__main__':
    install_shims()
    only = set(sys.argv[1:])
    sys.argv = sys.argv[:1]
    gens = dict(alias=gen_alias, bank=gen_bank, moco=gen_moco, dense=gen_dense, joint=gen_joint,
                scl=gen_scl, model=gen_model, model_bwd=gen_model_bwd, model_pn=gen_model_pn, pointnet2_msg=gen_pointnet2_msg, model_pn_fwd=gen_model_pn_fwd,
                model_pn_bwd=gen_model_pn_bwd, options=gen_options, trace=gen_trace,
                trace_moco=gen_trace_moco, dataset=gen_dataset)
    for name, fn in gens.items():
        if not only or name in only:
            fn()
