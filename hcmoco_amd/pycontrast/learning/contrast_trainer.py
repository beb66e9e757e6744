"""Contrastive pre-training loops (first stage: bank NCE; second stage: + dense, joint, SCL losses).

Host-side mirror of /root/reference/pycontrast/learning/contrast_trainer.py -- same public
methods (``wrap_up, broadcast_memory, resume_model, save, train, logging``), same checkpoint
layout and the same positional batch tuple -- around a loss engine whose
product implementation is the HIP kernels of this repo (learning/engine.py).

What changed relative to the reference, and why:
  * one packed all-gather per step ([B, 386] floats: features + bit-cast index) instead of two;
  * all three banks are broadcast at start-up (the reference forgets ``memory_3``);
  * meters stay on the device; the host syncs only every ``print_freq`` steps (reference: 13/step);
  * ``use_depth is None`` means "every sample has depth" and ``use_rgb is None`` "every sample has
    RGB" in the SCL loss (learning/segment_trainer.py:601-606); the reference crashes there.
"""
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from .base_trainer import BaseTrainer
from .engine import HipLossEngine
from .util import AverageMeter


class ContrastTrainer(BaseTrainer):
    def __init__(self, args, engine=None, force_collectives=None):
        super().__init__(args)
        self.engine = engine if engine is not None else HipLossEngine(getattr(args, 'fmap_dtype', 'fp32'))
        self.grad_sync = None        # learning/grad_sync.py:GradSync when this class averages the gradients itself
        self.async_wgrad = None      # torch.ops.hcmoco namespace once deferred weight gradients are on
        self._find_done = False      # the first training step runs single-stream (quiet MIOpen Find)
        # run every collective of the N>1 path even in a 1-rank group (GPU tests drive RCCL on a 1-GPU box)
        self.force_collectives = (os.environ.get('HCM_FORCE_COLLECTIVES', '0') != '0'
                                  if force_collectives is None else bool(force_collectives))

    def _multi(self):
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collectives)

    # ------------------------------------------------------------------ set-up / bookkeeping
    def logging(self, epoch, logs, lr):
        if self.args.rank != 0:
            return
        names = ('loss', 'acc', 'jig_loss', 'jig_acc')
        if self.logger is not None:
            for name, v in zip(names, logs):
                self.logger.log_value(name, v, epoch)
            self.logger.log_value('learning_rate', lr, epoch)
        else:
            print('epoch {} '.format(epoch) + ' '.join('{} {:.4f}'.format(n, v) for n, v in zip(names, logs)) +
                  ' learning_rate {:.6f}'.format(lr))

    def wrap_up(self, model, model_ema, optimizer):
        args = self.args
        model.to(self.device)
        # engines that can project at the sampled pixels get the raw branch maps from the model
        if (hasattr(self.engine, 'fmap_sampled') and hasattr(model, 'defer_projection')
                and getattr(args, 'sampled_projection', 1)):
            model.defer_projection = True
            # ... and, when it can, the heads too: the whole serial section becomes one autograd node
            if (getattr(self.engine, 'supports_section', None) is not None and hasattr(model, 'defer_heads')
                    and self.device.type == 'cuda' and getattr(args, 'fused_section', True)
                    and getattr(args, 'grad_sync', 'auto') != 'ddp' and not getattr(args, 'channels_last', False)
                    and self.engine.supports_section(model)):      # (the section kernels read NCHW branch maps)
                model.defer_heads = True
        if getattr(args, 'channels_last', False):
            # an r01 experiment for the stock-ATen encoders (HCMOCO_CHANNELS_LAST=1): the encoder runtime's kernels and
            # the loss section read NCHW
            from ..networks import hrnet as _hrnet
            if self.device.type == 'cuda' and (_hrnet.CONV_GLUE or _hrnet.ENCODER_PROGRAM or _hrnet.FUSED_BN):
                raise ValueError('channels_last needs the stock ATen encoders (networks.hrnet.CONV_GLUE / ENCODER_PROGRAM / '
                                 'FUSED_BN = False): the encoder runtime and the loss section are NCHW')
            model.to(memory_format=torch.channels_last)
        if isinstance(model_ema, torch.nn.Module):
            model_ema.to(self.device)
        # mixed precision (BASELINE config 5; the reference's hook is apex amp, contrast_trainer.py:65-72): the HRNets
        # run under bf16 autocast, everything else -- batch-norm statistics, master weights, SGD, the loss section --
        # stays fp32.  The model owns the switch (networks/build_backbone.py:_encode).
        if getattr(args, 'encoder_dtype', 'fp32') == 'bf16' or getattr(args, 'amp', False):
            if not hasattr(model, 'encoder_dtype'):
                raise NotImplementedError('--encoder_dtype bf16 / --amp needs a model with an encoder_dtype switch')
            model.encoder_dtype = torch.bfloat16
        multi = self._multi()
        sync = getattr(args, 'grad_sync', 'auto')
        if sync == 'auto':           # ROCm: own bucketed reduction (see below); CPU: DistributedDataParallel
            sync = 'overlap' if self.device.type == 'cuda' else 'ddp'
        deferred = (self.device.type == 'cuda' and sync != 'ddp'
                    and getattr(args, 'async_wgrad', True))      # False: the plain-autograd twin of the parity tests
        if deferred:
            # Deferred weight gradients (csrc/torch_glue): helper threads issue the encoders' reverse loops
            # and the MIOpen backward-weights calls while the autograd thread walks on.  The contract --
            # gradients reset with set_to_none, nothing reads .grad before wgrad_join() -- rules out
            # DistributedDataParallel's hooks (they fire when .grad is assigned, not when it is written),
            # so N > 1 uses learning/grad_sync.py: in-place RCCL all-reduces of the encoders' flat gradient
            # buffers, launched chunk by chunk while the reverse loops are still running.
            from ... import _lib
            self.async_wgrad = _lib.torch_glue()
            # r05: the weight gradients of the encoder programs on a side stream, 8 layers per hand-over: the
            # reverse chain of each encoder is what the step waits for, its dW kernels are not on it (same-box A/B:
            # HRNet x 2 704.7 / 705.4 -> 749.7 samples/s, HRNetPN 674 / 678 -> 707; batches of 16 / 32: 743 / 744)
            n = int(getattr(args, 'wgrad_stream', 8))
            self.async_wgrad.set_wgrad_stream(n > 0, max(n, 1))
        if multi and sync in ('flat', 'overlap'):
            from .grad_sync import GradSync, broadcast_model
            broadcast_model(model)
            glue = None
            if self.device.type == 'cuda':
                from ... import _lib
                glue = _lib.torch_glue()
            self.grad_sync = GradSync(model, [p for g in optimizer.param_groups for p in g['params']], mode=sync,
                                      chunks=int(os.environ.get('HCM_GRAD_CHUNKS', '4')), glue=glue)
        if deferred and getattr(args, 'flat_sgd', True) and isinstance(optimizer, torch.optim.SGD):
            # one update launch per encoder instead of ~40 per step (learning/flat_sgd.py); checkpoints keep the
            # reference's per-parameter layout
            from .flat_sgd import FlatParamSGD
            optimizer = FlatParamSGD(optimizer, model)
        if multi and self.grad_sync is None:
            ids = [self.device.index] if self.device.type == 'cuda' else None
            # stage 1 never touches the 1x1 feature-map projections when --linear_feat_map 1 is set
            unused = args.mem == 'bank' and bool(getattr(args, 'linear_feat_map', 0))
            model = DDP(model, device_ids=ids, gradient_as_bucket_view=True, find_unused_parameters=unused)
        if isinstance(model_ema, torch.nn.Module):
            self.momentum_update(self.unwrap(model), model_ema, 0)
        return model, model_ema, optimizer

    @staticmethod
    def unwrap(model):
        return model.module if isinstance(model, DDP) else model

    def broadcast_memory(self, contrast):
        """rank 0's banks everywhere -- all of them (contrast_trainer.py:81-91 skips memory_3)."""
        if not self._multi():
            return
        for name, buf in contrast.named_buffers():
            dist.broadcast(buf, 0)

    def _model_state(self, model):
        """'module.'-prefixed keys whether or not DDP wraps the model (main_contrast.py:57-58 and
        transfer_ckpt.py strip that 7-character prefix)."""
        sd = model.state_dict()
        return sd if isinstance(model, DDP) else {'module.' + k: v for k, v in sd.items()}

    def _load_model_state(self, model, sd):
        if not isinstance(model, DDP):
            sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        model.load_state_dict(sd)

    def resume_model(self, model, model_ema, contrast, optimizer):
        args = self.args
        start_epoch = 1
        if args.resume:
            if os.path.isfile(args.resume):
                ckpt = torch.load(args.resume, map_location='cpu')
                start_epoch = ckpt['epoch'] + 1
                self._load_model_state(model, ckpt['model'])
                contrast.load_state_dict(ckpt['contrast'])
                optimizer.load_state_dict(ckpt['optimizer'])
                if isinstance(model_ema, torch.nn.Module):
                    model_ema.load_state_dict(ckpt['model_ema'])
                sampler = getattr(contrast, 'multinomial', None)
                if sampler is not None and 'sampler' in ckpt:      # continue the negative stream, do not replay it
                    sampler.seed, sampler.offset = int(ckpt['sampler']['seed']), int(ckpt['sampler']['offset'])
                    if hasattr(contrast, '_pixel_draws'):       # the pixel sampler's own Philox stream
                        contrast._pixel_draws = int(ckpt['sampler'].get('pixel_draws', 0))
                print("=> resume successfully '{}' (epoch {})".format(args.resume, ckpt['epoch']))
                del ckpt
            else:
                print("=> no checkpoint found at '{}'".format(args.resume))
        return start_epoch

    def save(self, model, model_ema, contrast, optimizer, epoch):
        args = self.args
        if args.local_rank != 0:
            return
        print('==> Saving...')
        state = {'model': self._model_state(model), 'contrast': contrast.state_dict(),
                 'optimizer': optimizer.state_dict(), 'epoch': epoch}
        if isinstance(model_ema, torch.nn.Module):
            state['model_ema'] = model_ema.state_dict()
        sampler = getattr(contrast, 'multinomial', None)
        if sampler is not None and hasattr(sampler, 'offset'):
            # build-side key (the reference's sampler lives on torch's global generator, which it does not
            # checkpoint either): Philox (seed, offset) of the negative draws; loaders that index by key ignore it
            state['sampler'] = {'seed': sampler.seed, 'offset': sampler.offset,
                                'pixel_draws': int(getattr(contrast, '_pixel_draws', 0))}
        torch.save(state, os.path.join(args.model_folder, 'current.pth'))
        if epoch % args.save_freq == 0:
            torch.save(state, os.path.join(args.model_folder, 'ckpt_epoch_{}.pth'.format(epoch)))

    # ------------------------------------------------------------------ collectives
    @staticmethod
    def _global_gather(x):
        """reference helper (contrast_trainer.py:160-165), kept for API users."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return x
        out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(out, x.contiguous())
        return torch.cat(out, dim=0)

    @staticmethod
    def pack_features(f, index):
        """[B, D3] fp32 + [B] int64 -> [B, D3+2] fp32 (the index travels bit-cast in two floats)."""
        tail = index.to(torch.int64).contiguous().view(torch.int32).view(-1, 2).view(torch.float32)
        return torch.cat([f.detach().to(torch.float32), tail], dim=1).contiguous()

    @staticmethod
    def unpack_features(packed):
        f = packed[:, :-2]
        index = packed[:, -2:].contiguous().view(torch.int32).view(-1).view(torch.int64)
        return f, index

    @staticmethod
    def _gather_rows(packed):
        """[B, 3F+2] packed rows (written by the heads kernel) -> ([B*W, 3F+2] rank-major, work handle): the one
        collective of the forward pass, asynchronous -- the caller waits where it first reads the result."""
        out = torch.empty(dist.get_world_size() * packed.shape[0], packed.shape[1], dtype=packed.dtype,
                          device=packed.device)
        work = dist.all_gather_into_tensor(out, packed.contiguous(), async_op=True)
        return out, work

    def _packed_gather(self, f, index):
        """One collective per step; rank-major row order (it defines the duplicate-update winner)."""
        if not self._multi():
            return f.detach(), index
        mine = self.pack_features(f, index)
        out = torch.empty(dist.get_world_size() * mine.shape[0], mine.shape[1], dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine)
        return self.unpack_features(out)

    # ------------------------------------------------------------------ one training step
    def _to_dev(self, t):
        return t.to(self.device, non_blocking=True) if isinstance(t, torch.Tensor) else t

    def train_step(self, data, model, contrast, optimizer, stage2):
        """Forward, losses, backward, SGD step, bank update for one batch tuple.
        Returns a dict of DEVICE scalars (no host sync)."""
        if self.async_wgrad is not None and not self._find_done:
            # First step: MIOpen's Find benchmarks every convolution shape once.  Run it on a quiet
            # GPU -- one stream, nothing deferred -- so the timings it ranks algorithms by are not
            # disturbed by the other encoder; the records land in MIOpen's find-db, which the helper
            # threads' handles then hit instead of benchmarking under load.
            self._find_done = True
            net = self.unwrap(model)
            streams = getattr(net, 'two_streams', None)
            if streams is not None:
                net.two_streams = 0
            try:
                out = self._train_step(data, model, contrast, optimizer, stage2)
                torch.cuda.synchronize(self.device)
            finally:
                if streams is not None:
                    net.two_streams = streams
            return out
        if self.async_wgrad is None:
            return self._train_step(data, model, contrast, optimizer, stage2)
        # Deferred mode is process-global state of the encoder runtime: it is on only inside this step,
        # so any other backward in the process (linear probe, tests) sees complete .grad tensors.
        self.async_wgrad.set_async_wgrad(True)
        try:
            return self._train_step(data, model, contrast, optimizer, stage2)
        finally:
            self.async_wgrad.set_async_wgrad(False)         # joins the helper threads

    def _train_step(self, data, model, contrast, optimizer, stage2):
        args = self.args
        inputs = self._to_dev(data[0]).float()
        index = self._to_dev(data[1])
        skeleton = self._to_dev(data[2])
        use_depth = self._to_dev(data[6]) if args.modality_missing else None
        use_rgb = self._to_dev(data[11]) if len(data) > 11 else None
        if getattr(args, 'channels_last', False):
            inputs = inputs.contiguous(memory_format=torch.channels_last)

        if args.arch == 'HRNetPN':      # extra NTU items (appendix B rows 12-15; contrast_trainer.py:932-936)
            extra = (self._to_dev(data[7]), self._to_dev(data[12]), int(data[13][0]), int(data[14][0]),
                     self._to_dev(data[15]))
            if stage2:
                _feat1, _feat2, _feat3, f, aux = model(inputs, skeleton, *extra, return_fm=True)
            else:
                f = model(inputs, skeleton, *extra)
        elif stage2:
            _feat1, _feat2, _feat3, f, aux = model(inputs, skeleton, return_fm=True)
        else:
            f = model(inputs, skeleton)
        out = {}
        if stage2 and f is None:
            # the model ran with defer_heads: pooling, heads, all-gather, bank NCE + update, pixel sampling,
            # sampled projection and the three feature-map losses are ONE autograd node (engine.section)
            net = self.unwrap(model)
            if args.arch == 'HRNetPN':          # second modality: (cloud features, depth map at the HRNet's resolution)
                # (a view of the cloud features: the section's gradient w.r.t. its own input stays apart from the one
                # that reaches the same tensor through encoder2_linear -> the depth map -- what a recorder compares)
                _feat2 = (_feat2.view_as(_feat2), aux['linear_merge2'])
            loss, losses, accs, meters = self.engine.section(
                net, _feat1, _feat2, _feat3, index, contrast, True, depth_mask=self._to_dev(data[7]),
                joints2d=self._to_dev(data[4]), joints_vis=self._to_dev(data[5]), use_depth=use_depth, use_rgb=use_rgb,
                num_samples=args.pri3d_num_samples_per_image, temperature=args.temperature,
                gather=self._gather_rows if self._multi() else None)
            out['fmap'] = meters
            return self._finish_step(loss, losses, accs, optimizer, out)
        all_f, all_index = self._packed_gather(f, index)
        f1, f2, f3 = torch.chunk(f, 3, dim=1)
        all_f1, all_f2, all_f3 = torch.chunk(all_f, 3, dim=1)

        if stage2:      # stage 2 hands only use_depth to the bank CE (contrast_trainer.py:965-967)
            total, losses, accs = self.engine.bank(contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                                   use_depth=use_depth)
            fm_args = (_feat3, self._to_dev(data[7]), self._to_dev(data[4]), self._to_dev(data[5]), use_depth,
                       use_rgb, args.pri3d_num_samples_per_image, args.temperature)
            if aux['linear_merge1'] is None:     # model.defer_projection: project at the sampled pixels only
                net = self.unwrap(model)
                fm_total, meters = self.engine.fmap_sampled(_feat1, _feat2, net.encoder1_linear,
                                                            net.encoder2_linear, *fm_args)
            else:
                fm_total, meters = self.engine.fmap(aux['linear_merge1'], aux['linear_merge2'], *fm_args)
            loss = total + fm_total
            out['fmap'] = meters
        else:           # stage 1 (contrast_trainer.py:592-594)
            total, losses, accs = self.engine.bank(contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                                   use_depth=use_depth, use_rgb=use_rgb)
            loss = total
        return self._finish_step(loss, losses, accs, optimizer, out)

    def _finish_step(self, loss, losses, accs, optimizer, out):
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        join = self.async_wgrad.wgrad_join if self.async_wgrad is not None else None
        if self.grad_sync is not None:
            self.grad_sync.reduce(join)      # chunked RCCL all-reduces, launched while the helper threads still issue
        elif join is not None:
            join()
        optimizer.step()
        out.update(loss=loss.detach(), bank_losses=losses, bank_accs=accs)
        return out

    # ------------------------------------------------------------------ epoch loops
    def train(self, epoch, train_loader, model, model_ema, contrast, criterion, optimizer):
        args = self.args
        model.train()
        t0 = time.time()
        if args.mem == 'moco':
            outs = self._train_moco(epoch, train_loader, model, model_ema, contrast, criterion, optimizer)
        elif args.mem == 'bank':
            outs = self._train_mem_skeleton3d(epoch, train_loader, model, contrast, criterion, optimizer)
        elif args.mem == 'bank+jointspri3d':
            outs = self._train_bank_joints_pri3d_cmc3(epoch, train_loader, model, contrast, None, None, optimizer)
        else:
            raise NotImplementedError(args.mem)
        print('epoch {}, total time {:.2f}'.format(epoch, time.time() - t0))
        return outs

    def _run_epoch(self, epoch, train_loader, model, contrast, optimizer, stage2):
        args = self.args
        model.train()
        bt, dt = AverageMeter(), AverageMeter()
        loss_m = AverageMeter()
        acc_m = [AverageMeter() for _ in range(3)]
        pair_loss_m = [AverageMeter() for _ in range(3)]
        fmap_m = [AverageMeter() for _ in range(9)]
        end = time.time()
        nb = len(train_loader)
        for idx, data in enumerate(train_loader):
            dt.update(time.time() - end)
            bsz = data[0].size(0)
            self.warmup_learning_rate(epoch, idx, nb, optimizer)
            out = self.train_step(data, model, contrast, optimizer, stage2)
            loss_m.update(out['loss'], bsz)
            for k in range(3):   # pairs (12,21) (23,32) (13,31) averaged like the reference (:982-987)
                acc_m[k].update(0.5 * (out['bank_accs'][2 * k] + out['bank_accs'][2 * k + 1]), bsz)
                pair_loss_m[k].update(0.5 * (out['bank_losses'][2 * k] + out['bank_losses'][2 * k + 1]), bsz)
            if stage2:
                for k in range(9):
                    fmap_m[k].update(out['fmap'][k], bsz)
            bt.update(time.time() - end)
            end = time.time()
            if (idx + 1) % args.print_freq == 0 or idx + 1 == nb:
                if hasattr(contrast, 'check_indices'):
                    contrast.check_indices()                    # a sync point anyway (meters are printed here)
            if args.local_rank == 0 and (idx + 1) % args.print_freq == 0:
                msg = ('Train: [{0}][{1}/{2}]\tBT {3:.3f} ({4:.3f})\tDT {5:.3f} ({6:.3f})\tL {7:.3f} ({8:.3f})\t'
                       'a_I {9:.3f} {10:.3f} {11:.3f}').format(epoch, idx + 1, nb, bt.val, bt.avg, dt.val, dt.avg,
                                                            loss_m.val, loss_m.avg, acc_m[0].avg, acc_m[1].avg,
                                                            acc_m[2].avg)
                if stage2:
                    f = [m.avg for m in fmap_m]
                    msg += ('\tp3d {0:.3f} {2:.3f} {1:.3f} {3:.3f}\tj {4:.3f} {6:.3f} {5:.3f} {7:.3f}\tscl {8:.3f}'
                            .format(*f))
                print(msg)
                sys.stdout.flush()
        return loss_m, acc_m, pair_loss_m

    def _train_mem_skeleton3d(self, epoch, train_loader, model, contrast, criterion, optimizer):
        """stage 1 (contrast_trainer.py:532-640): returns the six per-pair meters."""
        assert not self.args.jigsaw
        _, acc_m, pl = self._run_epoch(epoch, train_loader, model, contrast, optimizer, stage2=False)
        return pl[0].avg, acc_m[0].avg, pl[1].avg, acc_m[1].avg, pl[2].avg, acc_m[2].avg

    def _train_bank_joints_pri3d_cmc3(self, epoch, train_loader, model, contrast, criterion_contrast,
                                      criterion_pri3d, optimizer):
        """stage 2 (contrast_trainer.py:894-1039): (loss, acc12, jig_loss=0, jig_acc=0)."""
        loss_m, acc_m, _ = self._run_epoch(epoch, train_loader, model, contrast, optimizer, stage2=True)
        return loss_m.avg, acc_m[0].avg, 0.0, 0.0

    # ------------------------------------------------------------------ MoCo (secondary row of SURVEY 8a)
    @staticmethod
    def _ce_accuracy(logits, target):
        """nn.CrossEntropyLoss + top-1 accuracy in percent per logit set: the plain branch of
        _compute_loss_accuracy (contrast_trainer.py:212-222, :249-251; learning/util.py:24-38)."""
        losses = [torch.nn.functional.cross_entropy(l, target) for l in logits]
        accs = [(l.argmax(dim=1) == target).float().sum() * (100.0 / target.shape[0]) for l in logits]
        return losses, accs

    def _shuffle_bn(self, x, model_ema, shuffle_ids=None):
        """Shuffle-BN forward of the momentum encoder (contrast_trainer.py:167-210): the key crops of one node are
        gathered, permuted with a permutation rank 0 broadcasts, every GPU encodes a shuffled slice (so the key
        encoder's batch statistics do not leak which keys belong to this GPU's queries), the keys are gathered from
        everybody and un-shuffled.  -> (k for the local batch, all_k in rank-major order).  One collective per gather
        (``all_gather_into_tensor``); with a single process the shuffle is a permutation of the local batch.
        ``shuffle_ids`` injects the permutation (parity tests); otherwise ``torch.randperm`` on the host generator,
        as in the reference."""
        args = self.args
        bsz = x.size(0)
        multi = dist.is_initialized() and dist.get_world_size() > 1
        nlocal = dist.get_world_size(self.local_group) if multi and self.local_group is not None else \
            (dist.get_world_size() if multi else 1)
        if multi:
            node_x = torch.empty((nlocal * bsz,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(node_x, x.contiguous(), group=self.local_group)
        else:
            node_x = x
        if shuffle_ids is None:
            shuffle_ids = torch.randperm(bsz * nlocal)
        shuffle_ids = shuffle_ids.to(x.device)
        if multi:                                       # rank 0's permutation for everybody (:184-187)
            dist.broadcast(shuffle_ids, 0)
        reverse_ids = torch.argsort(shuffle_ids)
        # the slice this rank encodes = its position INSIDE the group the crops were gathered over (not torchrun's
        # LOCAL_RANK: with fewer ranks than GPUs per node, or on gloo, the two differ -- ADVICE r03)
        if multi:
            local = dist.get_rank(self.local_group) if self.local_group is not None else dist.get_rank()
        else:
            local = 0
        this_ids = shuffle_ids[local * bsz:(local + 1) * bsz]
        with torch.no_grad():
            k = model_ema(node_x[this_ids], mode=1)
        if multi:
            all_k = torch.empty((dist.get_world_size() * bsz,) + tuple(k.shape[1:]), dtype=k.dtype, device=k.device)
            dist.all_gather_into_tensor(all_k, k.contiguous())
            node = (dist.get_rank() // nlocal) if self.local_group is not None else 0      # groups are contiguous rank blocks
            node_k = all_k[node * nlocal * bsz:(node + 1) * nlocal * bsz]
        else:
            all_k = node_k = k
        k = node_k[reverse_ids[local * bsz:(local + 1) * bsz]]
        return k, all_k

    def _train_moco(self, epoch, train_loader, model, model_ema, contrast, criterion, optimizer):
        """MoCo-style epoch (contrast_trainer.py:255-389) for the RGB and CMC modalities without jigsaw: query encoder
        on crop 1, shuffle-BN momentum encoder on crop 2, ``contrast`` = RGBMoCo / CMCMoCo (memory/mem_moco.py: logits
        against the ring queue on the HIP kernels, then the enqueue), cross entropy, SGD step, momentum update of the
        key encoder.  Returns (loss, acc, jig_loss, jig_acc) like the reference."""
        args = self.args
        if getattr(args, 'jigsaw', False):
            raise NotImplementedError('the jigsaw branch of the MoCo loop needs the JigsawHead models (out of scope)')
        model.train()
        model_ema.eval()
        for mod in model_ema.modules():                 # BN of the key encoder stays in training mode (:266-270)
            if 'BatchNorm' in mod.__class__.__name__:
                mod.train()
        loss_m, acc_m, bt = AverageMeter(), AverageMeter(), AverageMeter()
        end = time.time()
        nb = len(train_loader)
        inject = getattr(self, 'inject_shuffle_ids', None)       # parity tests: the reference's recorded permutations
        for idx, data in enumerate(train_loader):
            inputs = self._to_dev(data[0]).float()
            bsz = inputs.size(0)
            self.warmup_learning_rate(epoch, idx, nb, optimizer)
            x1, x2 = torch.split(inputs, [3, 3], dim=1)
            k, all_k = self._shuffle_bn(x2, model_ema, None if inject is None else inject[idx])
            q = model(x1)
            if args.modal == 'CMC':
                q1, q2 = torch.chunk(q, 2, dim=1)
                k1, k2 = torch.chunk(k, 2, dim=1)
                all_k1, all_k2 = torch.chunk(all_k, 2, dim=1)
                output = contrast(q1.contiguous(), k1.contiguous(), q2.contiguous(), k2.contiguous(),
                                  all_k1=all_k1.contiguous(), all_k2=all_k2.contiguous())
                losses, accs = self._ce_accuracy(output[:-1], output[-1])
                loss = losses[0] + losses[1]
                update_loss, update_acc = 0.5 * (losses[0] + losses[1]), 0.5 * (accs[0] + accs[1])
            else:
                output = contrast(q, k, all_k=all_k)
                losses, accs = self._ce_accuracy(output[:-1], output[-1])
                loss = update_loss = losses[0]
                update_acc = accs[0]
            self.last_moco = {'logits': [l.detach() for l in output[:-1]], 'losses': [l.detach() for l in losses],
                              'accs': accs}
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            loss_m.update(update_loss.detach(), bsz)
            acc_m.update(update_acc, bsz)
            self.momentum_update(self.unwrap(model), model_ema, args.alpha)
            bt.update(time.time() - end)
            end = time.time()
            if args.local_rank == 0 and (idx + 1) % args.print_freq == 0:
                print('Train: [{0}][{1}/{2}]\tBT {3:.3f} ({4:.3f})\tl_I {5:.3f} ({6:.3f})\ta_I {7:.3f} ({8:.3f})'
                      .format(epoch, idx + 1, nb, bt.val, bt.avg, float(loss_m.val), float(loss_m.avg),
                              float(acc_m.val), float(acc_m.avg)))
                sys.stdout.flush()
        return float(loss_m.avg), float(acc_m.avg), 0.0, 0.0

    @staticmethod
    def momentum_update(model, model_ema, m):
        """model_ema = m * model_ema + (1 - m) * model (contrast_trainer.py:1041-1045)."""
        with torch.no_grad():
            for p1, p2 in zip(model.parameters(), model_ema.parameters()):
                p2.mul_(m).add_(p1.detach(), alpha=1 - m)
