"""Loss engine of the contrastive trainer: the seam between host code and the HIP kernels.

``HipLossEngine`` is the product path: every method lands in ``libhcmoco_hip.so`` through
``hcmoco_amd.hip_ops`` and raises on CPU tensors / a missing library.  The trainer only talks to
this interface, which is what lets the CPU test-suite drive the same training loop with an
oracle-backed engine (oracle/oracle_engine.py) without the product ever importing the oracle.
"""
import torch
import torch.nn.functional as F

from ... import hip_ops


class HipLossEngine(object):
    name = 'hip'

    def __init__(self, fmap_dtype='fp32'):
        """``fmap_dtype='bf16'`` (``--fmap_dtype bf16``, BASELINE config 5): the dense and SCL similarity /
        gradient contractions run on the bf16 matrix cores with fp32 accumulation."""
        if fmap_dtype not in ('fp32', 'bf16', 'fp32_exact'):
            raise ValueError('fmap_dtype must be fp32, bf16 or fp32_exact')
        self.fmap_dtype = fmap_dtype
        # fp32_exact means exact fp32 contractions wherever the default is split-bf16: the SharedMLP 1x1 convolutions of the
        # PointNet++ branch too (hcm_conv1x1_set_arith, process-wide)
        from ... import pointnet2_hip
        pointnet2_hip.set_conv1x1_arith(exact=(fmap_dtype == 'fp32_exact'))

    # ---- SURVEY 8a rows 1-4 -------------------------------------------------------------
    def bank(self, contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
             use_depth=None, use_rgb=None, idx=None):
        """-> (total, losses[6], accs[6]); updates the banks in place after the reads."""
        return contrast.forward_loss(f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                     use_depth=use_depth, use_rgb=use_rgb, idx=idx)

    # ---- sampling shared by rows 5-7 (host-side torch glue, stream ordered, no sync) ----------
    @staticmethod
    def dense_samples(depth_mask, h, w, num_samples, use_depth=None, generator=None):
        """Pixel sampling of _compute_soft_pri3d_loss_accuracy (contrast_trainer.py:671-685):
        nearest-resize the mask, keep images whose resized mask is non-empty, draw S pixels per
        kept image with replacement.  Dropped images get placeholder indices and keep=0."""
        m = F.interpolate(depth_mask.unsqueeze(1).float(), size=(h, w), mode='nearest').reshape(depth_mask.shape[0], h * w)
        keep = m.sum(-1) > 0
        if use_depth is not None:                       # reference early return (:663-665)
            keep = keep & (use_depth.sum() > 0)
        weights = m + (~keep).unsqueeze(1).to(m.dtype)  # all-ones rows for dropped images: never read
        ind = torch.multinomial(weights, num_samples, replacement=True, generator=generator)
        return ind, keep.to(torch.int32)

    # ---- SURVEY 8a rows 5-7 -------------------------------------------------------------
    def fmap(self, map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb,
             num_samples, temperature, sample_ind=None, keep=None):
        """-> (total, meters[9]) for the dense, joint and SCL losses (contrast_trainer.py:969-980).
        meters = [loss_r2d, loss_d2r, acc_r2d, acc_d2r, loss_rgb2j, loss_d2j, acc_rgb2j, acc_d2j, loss_scl]."""
        B, C, h, w = map1.shape
        assert h == w                                   # contrast_trainer.py:751
        if sample_ind is None:
            sample_ind, keep = self.dense_samples(depth_mask, h, w, num_samples, use_depth)
        pix = hip_ops.joint_pixels(joints2d, h)
        ud = use_depth if use_depth is not None else torch.ones(B, dtype=torch.int32, device=map1.device)
        return hip_ops.fmap_losses(map1, map2, feat3, sample_ind, keep, pix, joints_vis, ud, use_rgb, temperature,
                                   gemm_dtype=self.fmap_dtype)

    # ---- rows 5-8 fused at the sampled pixels ---------------------------------
    def fmap_sampled(self, branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                     use_depth, use_rgb, num_samples, temperature, sample_ind=None, keep=None):
        """Same losses as ``fmap`` computed from the raw HRNet branch maps: the 1x1 projections
        ``proj1/proj2`` (``encoder{1,2}_linear``) are applied only at the S+J pixels the losses read
        (``hip_ops.sampled_projection``), so ``merge_all_res`` and the full-resolution projection
        are never materialised."""
        h, w = branches1[0].shape[-2:]
        assert h == w
        B = branches1[0].shape[0]
        if sample_ind is None:
            sample_ind, keep = self.dense_samples(depth_mask, h, w, num_samples, use_depth)
        pix_j = hip_ops.joint_pixels(joints2d, h)
        pix = torch.cat([sample_ind, pix_j], dim=1).contiguous()
        S = [hip_ops.sampling_matrix(pix, m.shape[2], m.shape[3], h, w) for m in branches1[1:]]   # shared
        rows1 = hip_ops.sampled_projection(proj1.weight, proj1.bias, pix, list(branches1), S)
        rows2 = hip_ops.sampled_projection(proj2.weight, proj2.bias, pix, list(branches2), S)
        ud = use_depth if use_depth is not None else torch.ones(B, dtype=torch.int32, device=pix.device)
        return hip_ops.fmap_losses_rows(rows1, rows2, feat3, sample_ind.shape[1], sample_ind, w, keep, joints_vis,
                                        ud, use_rgb, temperature, gemm_dtype=self.fmap_dtype)

    # ---- rows 1-9 of a step as one autograd node ------------------------------
    @staticmethod
    def supports_section(net, stage2=True):
        """The fused section covers the RGBD2S HRNet model -- and, for stage 2, the HRNetPN model (r05) -- with mean pooling
        and linear heads."""
        kind = type(net).__name__
        if kind == 'CMC3HRNetSGCNPN2SingleHead':
            ok = stage2 and bool(net.linear_feat_map) and net.head2[0].weight.shape[1] <= net.head1[0].weight.shape[1]
        else:
            ok = kind == 'CMC3HRNetSGCNSingleHead' and (not stage2 or bool(net.linear_feat_map))
        return (ok and net.pool_method == 'mean' and net.head1[0].weight.shape[0] <= 256 and net.head1[0].weight.is_cuda)

    def section(self, net, branches1, branches2, feat3, index, contrast, stage2, depth_mask=None, joints2d=None,
                joints_vis=None, use_depth=None, use_rgb=None, num_samples=0, temperature=0.07, gather=None,
                idx=None, sample_ind=None, keep=None, tape=None):
        """Everything between the encoders' forward and backward as ONE autograd node (``hip_ops.stage2_section``,
        csrc/section.hip): heads (build_backbone.py:265-288) -> [``gather``: the packed feature/index all-gather,
        contrast_trainer.py:950-951] -> negative draw + fused bank NCE + momentum update (mem_bank.py:172-205) ->
        pixel sampling (contrast_trainer.py:671-685) -> merge_all_res + projection at the sampled pixels
        (build_backbone.py:243-254) -> dense / joint / SCL losses (:642-892), ~35 launches.  The model must have been
        run with ``defer_heads`` (it returns the raw branch maps and no ``f``).
        -> (total, bank_losses[6], bank_accs[6], meters[9])."""
        heads = (net.head1[0].weight, net.head1[0].bias, net.head2[0].weight, net.head2[0].bias,
                 net.head3[0].weight, net.head3[0].bias)
        if type(net).__name__ == 'CMC3HRNetSGCNPN2SingleHead':
            # HRNetPN: ``branches2`` = (cloud features [B, C2, Npts], depth map [B, F, h, w]) -- hip_ops.stage2_section_pn
            feat2, lm2 = branches2
            h, w = branches1[0].shape[-2:]
            assert h == w and stage2
            cfg = dict(contrast=contrast, index=index, use_depth=use_depth, use_rgb=use_rgb, depth_mask=depth_mask,
                       joints2d=joints2d, joints_vis=joints_vis, num_samples=num_samples, temperature=temperature,
                       gemm_dtype=self.fmap_dtype, gather=gather, idx=idx, sample_ind=sample_ind, keep=keep, tape=tape)
            return hip_ops.stage2_section_pn(feat3, heads, (net.encoder1_linear.weight, net.encoder1_linear.bias),
                                             list(branches1), feat2, lm2, cfg)
        if stage2:
            projs = (net.encoder1_linear.weight, net.encoder1_linear.bias, net.encoder2_linear.weight,
                     net.encoder2_linear.bias)
            h, w = branches1[0].shape[-2:]
            assert h == w                               # contrast_trainer.py:751
        else:
            projs = (None, None, None, None)
        cfg = dict(contrast=contrast, index=index, use_depth=use_depth, use_rgb=use_rgb, depth_mask=depth_mask,
                   joints2d=joints2d, joints_vis=joints_vis, num_samples=num_samples, temperature=temperature,
                   gemm_dtype=self.fmap_dtype, gather=gather, idx=idx, sample_ind=sample_ind, keep=keep, tape=tape,
                   stage2=bool(stage2), bank_use_rgb=not stage2)
        return hip_ops.stage2_section(feat3, heads, projs, list(branches1), list(branches2), cfg)


class RecordingEngine(object):
    """Wrapper around a loss engine (the product's ``HipLossEngine`` unless told otherwise) that keeps CPU copies of everything one training step hands to the loss kernels and
    everything they hand back -- bank rows before the update, the negative indices, sampled pixels, features,
    feature maps / branch maps, projection weights, losses, accuracies, the updated bank rows and (through
    autograd hooks) the gradients -- so that an EXTERNAL checker can re-evaluate the step
    (``oracle/check_step.py``: the ``-m gpu`` whole-step tests and the ``--check`` leg of bench.py, which runs the
    checker in a separate CPU process).  It computes nothing itself and imports nothing outside the product: the
    numbers recorded are the HIP path's own.  Random draws are made by the same product code one call earlier
    (``draw_with_positive`` / ``dense_samples``) and handed in through the ``idx=`` / ``sample_ind=`` arguments."""
    name = 'hip+record'
    dense_samples = staticmethod(HipLossEngine.dense_samples)

    def __init__(self, fmap_dtype='fp32', inner=None):
        self.inner = inner if inner is not None else HipLossEngine(fmap_dtype)
        self.fmap_dtype = fmap_dtype
        self.records = []
        self.armed = True

    @staticmethod
    def _cpu(t):
        if t is None:
            return None
        if isinstance(t, (list, tuple)):
            return [RecordingEngine._cpu(v) for v in t]
        return t.detach().to('cpu', copy=True)

    def _grads(self, rec, total, named):
        """d total / d t for every named tensor that takes part in autograd, through the product's own backward
        kernels (``retain_graph``: the trainer's ``loss.backward()`` afterwards is undisturbed)."""
        named = [(k, t) for k, t in named if t is not None and t.requires_grad]
        if not named or not total.requires_grad:
            return
        gs = torch.autograd.grad(total, [t for _, t in named], retain_graph=True, allow_unused=True)
        for (k, _), g in zip(named, gs):
            rec['grads'][k] = None if g is None else g.detach().cpu()

    def bank(self, contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
             use_depth=None, use_rgb=None, idx=None):
        if not self.armed:
            return self.inner.bank(contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                use_depth=use_depth, use_rgb=use_rgb, idx=idx)
        c = self._cpu
        if idx is None:
            draw = getattr(self.inner, 'draw_indices', None)      # engines that cannot run the device sampler
            idx = draw(contrast, index) if draw is not None else contrast.multinomial.draw_with_positive(index, contrast.K + 1)
        before = [b.detach().clone() for b in contrast.banks()]
        rec = {'kind': 'bank', 'T': contrast.T, 'm': contrast.m, 'banks0': c(before), 'idx': c(idx),
               'x': c([f1, f2, f3]), 'index': c(index), 'all_x': c([all_f1, all_f2, all_f3]),
               'all_index': c(all_index), 'use_depth': c(use_depth), 'use_rgb': c(use_rgb), 'grads': {}}
        total, losses, accs = self.inner.bank(contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                           use_depth=use_depth, use_rgb=use_rgb, idx=idx)
        rec['losses'], rec['accs'], rec['total'] = c(losses), c(accs), c(total)
        touched = torch.zeros(before[0].shape[0], dtype=torch.bool, device=before[0].device)
        touched[all_index.clamp(0, before[0].shape[0] - 1)] = True
        rec['after_rows'] = [c(b.index_select(0, all_index)) for b in contrast.banks()]
        rec['untouched_rows_unchanged'] = [bool(torch.equal(b[~touched], b0[~touched]))
                                           for b, b0 in zip(contrast.banks(), before)]
        self._grads(rec, total, [('x%d' % (i + 1), f) for i, f in enumerate((f1, f2, f3))])
        self.records.append(rec)
        return total, losses, accs

    def fmap(self, map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb,
             num_samples, temperature, sample_ind=None, keep=None):
        if not self.armed:
            return self.inner.fmap(map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb,
                                num_samples, temperature, sample_ind=sample_ind, keep=keep)
        c = self._cpu
        h, w = map1.shape[-2:]
        if sample_ind is None:
            sample_ind, keep = self.dense_samples(depth_mask, h, w, num_samples, use_depth)
        rec = {'kind': 'fmap', 'map1': c(map1), 'map2': c(map2), 'feat3': c(feat3), 'depth_mask': c(depth_mask),
               'joints2d': c(joints2d), 'joints_vis': c(joints_vis), 'use_depth': c(use_depth), 'use_rgb': c(use_rgb),
               'num_samples': int(num_samples), 'temperature': float(temperature), 'sample_ind': c(sample_ind),
               'keep': c(keep), 'fmap_dtype': self.fmap_dtype, 'grads': {}}
        total, meters = self.inner.fmap(map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb,
                                     num_samples, temperature, sample_ind=sample_ind, keep=keep)
        rec['meters'], rec['total'] = c(meters), c(total)
        self._grads(rec, total, [('map1', map1), ('map2', map2), ('feat3', feat3)])
        self.records.append(rec)
        return total, meters

    def fmap_sampled(self, branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                     use_depth, use_rgb, num_samples, temperature, sample_ind=None, keep=None):
        if not self.armed:
            return self.inner.fmap_sampled(branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                                        use_depth, use_rgb, num_samples, temperature, sample_ind=sample_ind, keep=keep)
        c = self._cpu
        h, w = branches1[0].shape[-2:]
        if sample_ind is None:
            sample_ind, keep = self.dense_samples(depth_mask, h, w, num_samples, use_depth)
        rec = {'kind': 'fmap_sampled', 'branches1': c(list(branches1)), 'branches2': c(list(branches2)),
               'proj1': (c(proj1.weight), c(proj1.bias)), 'proj2': (c(proj2.weight), c(proj2.bias)),
               'feat3': c(feat3), 'depth_mask': c(depth_mask), 'joints2d': c(joints2d), 'joints_vis': c(joints_vis),
               'use_depth': c(use_depth), 'use_rgb': c(use_rgb), 'num_samples': int(num_samples),
               'temperature': float(temperature), 'sample_ind': c(sample_ind), 'keep': c(keep),
               'fmap_dtype': self.fmap_dtype, 'grads': {}}
        total, meters = self.inner.fmap_sampled(branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d,
                                             joints_vis, use_depth, use_rgb, num_samples, temperature,
                                             sample_ind=sample_ind, keep=keep)
        rec['meters'], rec['total'] = c(meters), c(total)
        named = [('b1_%d' % i, t) for i, t in enumerate(branches1)] + [('b2_%d' % i, t) for i, t in enumerate(branches2)]
        named += [('feat3', feat3), ('proj1_w', proj1.weight), ('proj1_b', proj1.bias), ('proj2_w', proj2.weight),
                  ('proj2_b', proj2.bias)]
        self._grads(rec, total, named)
        self.records.append(rec)
        return total, meters

    def supports_section(self, net, stage2=True):
        fn = getattr(self.inner, 'supports_section', None)
        return bool(fn is not None and fn(net, stage2))

    def section(self, net, branches1, branches2, feat3, index, contrast, stage2, depth_mask=None, joints2d=None,
                joints_vis=None, use_depth=None, use_rgb=None, num_samples=0, temperature=0.07, gather=None,
                idx=None, sample_ind=None, keep=None, tape=None):
        kw = dict(depth_mask=depth_mask, joints2d=joints2d, joints_vis=joints_vis, use_depth=use_depth, use_rgb=use_rgb,
                  num_samples=num_samples, temperature=temperature, gather=gather, idx=idx, sample_ind=sample_ind,
                  keep=keep)
        if not self.armed:
            return self.inner.section(net, branches1, branches2, feat3, index, contrast, stage2, tape=tape, **kw)
        c = self._cpu
        tape = {} if tape is None else tape
        heads = [net.head1[0], net.head2[0], net.head3[0]]
        pn = type(net).__name__ == 'CMC3HRNetSGCNPN2SingleHead'
        projs = ([net.encoder1_linear] if pn else [net.encoder1_linear, net.encoder2_linear]) if stage2 else []
        rec = {'kind': 'section_pn' if pn else 'section', 'stage2': bool(stage2), 'T': contrast.T, 'm': contrast.m,
               'branches1': c(list(branches1)), 'branches2': c(list(branches2)), 'feat3': c(feat3),
               'heads': [(c(l.weight), c(l.bias)) for l in heads], 'projs': [(c(l.weight), c(l.bias)) for l in projs],
               'index': c(index), 'depth_mask': c(depth_mask), 'joints2d': c(joints2d), 'joints_vis': c(joints_vis),
               'use_depth': c(use_depth), 'use_rgb': c(use_rgb), 'num_samples': int(num_samples),
               'temperature': float(temperature), 'fmap_dtype': self.fmap_dtype, 'grads': {}}
        total, losses, accs, meters = self.inner.section(net, branches1, branches2, feat3, index, contrast, stage2,
                                                         tape=tape, **kw)
        rec.update(total=c(total), losses=c(losses), accs=c(accs), meters=c(meters), banks0=c(tape['banks0']),
                   idx=c(tape['idx']), f=c(tape['f']), all_x=c(tape['all_x']), all_index=c(tape['all_index']))
        if stage2:
            S = int(num_samples)
            rec.update(sample_ind=c(tape['coord']), keep=c(tape['keep']), pix=c(tape['pix']))
        banks = contrast.banks()
        ai = tape['all_index'].clamp(0, banks[0].shape[0] - 1)
        touched = torch.zeros(banks[0].shape[0], dtype=torch.bool, device=banks[0].device)
        touched[ai] = True
        rec['after_rows'] = [c(b.index_select(0, ai)) for b in banks]
        rec['untouched_rows_unchanged'] = [bool(torch.equal(b[~touched], b0[~touched]))
                                           for b, b0 in zip(banks, tape['banks0'])]
        named = [('b1_%d' % i, t) for i, t in enumerate(branches1)]
        named += [('feat2', branches2[0]), ('lm2', branches2[1])] if pn else [('b2_%d' % i, t) for i, t in enumerate(branches2)]
        named += [('feat3', feat3)]
        named += [('head%d_%s' % (i + 1, n), getattr(l, a)) for i, l in enumerate(heads) for n, a in (('w', 'weight'), ('b', 'bias'))]
        named += [('proj%d_%s' % (i + 1, n), getattr(l, a)) for i, l in enumerate(projs) for n, a in (('w', 'weight'), ('b', 'bias'))]
        self._grads(rec, total, named)
        self.records.append(rec)
        return total, losses, accs, meters
