"""Oracle-backed loss engine (TEST INFRASTRUCTURE).  Implements the interface of
``hcmoco_amd.pycontrast.learning.engine.HipLossEngine`` on CPU tensors with oracle/hcmoco_oracle.py
so the CPU suite, smoke() and the cpu_baseline leg of bench.py can drive the real trainer loop
without a GPU.  The product package never imports this."""
import torch

from oracle import hcmoco_oracle as O
from hcmoco_amd.pycontrast.learning.engine import HipLossEngine


class _Bank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, x3, banks, idx, T, use_depth, use_rgb):
        losses, accs, grads, _ = O.bank_nce(banks, idx, [x1, x2, x3], T, use_depth, use_rgb)
        ctx.save_for_backward(*grads)
        return losses.sum(), losses, accs

    @staticmethod
    def backward(ctx, g, *_):
        return tuple(gr * g for gr in ctx.saved_tensors) + (None,) * 5


class _Fmap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, map1, map2, feat3, sample_ind, keep, joints2d, vis, use_depth, use_rgb, temperature):
        keepb = keep.bool()
        ld, ad, d1, d2 = O.dense_soft_nce(map1, map2, sample_ind[keepb], keepb, temperature,
                                          None if bool(keepb.any()) else torch.zeros(1))
        lj, aj, j1, j2, j3 = O.joint_nce(map1, map2, feat3, joints2d, vis, temperature, use_depth)
        ud = use_depth if use_depth is not None else torch.ones(map1.shape[0])
        ls, s1, s2, _ = O.scl(map1, map2, joints2d, temperature, ud, use_rgb)
        ctx.save_for_backward(d1 + j1 + s1, d2 + j2 + s2, j3)
        meters = torch.cat([ld, ad, lj, aj, ls.reshape(1)])
        return ld.sum() + lj.sum() + ls, meters

    @staticmethod
    def backward(ctx, g, _):
        g1, g2, g3 = ctx.saved_tensors
        return (g1 * g, g2 * g, g3 * g) + (None,) * 7


class OracleLossEngine(object):
    name = 'oracle'

    @staticmethod
    def draw_indices(contrast, index):
        """idx [B, K+1] with idx[:,0] = index: the product's Philox draw restated on the host."""
        mn = contrast.multinomial
        B, K1 = index.shape[0], contrast.K + 1
        idx = O.alias_draw_philox(mn.prob, mn.alias, B * K1, mn.seed, mn.offset).view(B, K1).clone()
        idx[:, 0] = index
        mn.offset += 1
        return idx

    def bank(self, contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
             use_depth=None, use_rgb=None, idx=None):
        if idx is None:
            idx = self.draw_indices(contrast, index)
        banks = [b.clone() for b in contrast.banks()]
        total, losses, accs = _Bank.apply(f1, f2, f3, banks, idx, contrast.T, use_depth, use_rgb)
        with torch.no_grad():
            for bank, ax in zip(contrast.banks(), (all_f1, all_f2, all_f3)):
                bank.copy_(O.bank_update(bank, ax.detach(), all_index, contrast.m))
        return total, losses.detach(), accs.detach()

    def fmap(self, map1, map2, feat3, depth_mask, joints2d, joints_vis, use_depth, use_rgb,
             num_samples, temperature, sample_ind=None, keep=None):
        B, C, h, w = map1.shape
        if sample_ind is None:
            sample_ind, keep = HipLossEngine.dense_samples(depth_mask, h, w, num_samples, use_depth)
        total, meters = _Fmap.apply(map1, map2, feat3, sample_ind, keep, joints2d, joints_vis, use_depth, use_rgb,
                                    temperature)
        return total, meters.detach()


    def fmap_sampled(self, branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                     use_depth, use_rgb, num_samples, temperature, sample_ind=None, keep=None):
        """CPU counterpart of HipLossEngine.fmap_sampled: the reference data flow in plain torch
        (merge_all_res + 1x1 projection of the full maps) followed by the oracle losses.  The
        projection weights are used OUTSIDE model.forward here, exactly like in the product path,
        which is what the world_size-2 test needs to exercise under DistributedDataParallel."""
        import torch.nn.functional as F

        def project(branches, conv):
            size = branches[0].shape[-2:]
            up = [branches[0]] + [F.interpolate(m, size=size, mode='bilinear', align_corners=False) for m in branches[1:]]
            return conv(torch.cat(up, 1))
        return self.fmap(project(branches1, proj1), project(branches2, proj2), feat3, depth_mask, joints2d,
                         joints_vis, use_depth, use_rgb, num_samples, temperature, sample_ind, keep)


class OracleCMCMoCo(torch.nn.Module):
    """CPU counterpart of ``hcmoco_amd.pycontrast.memory.mem_moco.CMCMoCo`` on the oracle's queue functions
    (memory/mem_moco.py:91-142 of the reference): lets the CPU suite drive the trainer's MoCo loop."""

    def __init__(self, n_dim, K=65536, T=0.07):
        super().__init__()
        self.K, self.T, self.index = K, T, 0
        self.register_buffer('memory_1', torch.nn.functional.normalize(torch.randn(K, n_dim)))
        self.register_buffer('memory_2', torch.nn.functional.normalize(torch.randn(K, n_dim)))

    def forward(self, q1, k1, q2, k2, q1_jig=None, q2_jig=None, all_k1=None, all_k2=None):
        k1, k2 = k1.detach(), k2.detach()
        logits1 = O.moco_logits(q1, k2, self.memory_2.clone(), self.T)
        logits2 = O.moco_logits(q2, k1, self.memory_1.clone(), self.T)
        labels = torch.zeros(q1.shape[0], dtype=torch.long)
        all_k1 = all_k1 if all_k1 is not None else k1
        all_k2 = all_k2 if all_k2 is not None else k2
        with torch.no_grad():
            m1, nxt = O.moco_enqueue(self.memory_1, all_k1.detach(), self.index)
            m2, _ = O.moco_enqueue(self.memory_2, all_k2.detach(), self.index)
            self.memory_1.copy_(m1)
            self.memory_2.copy_(m2)
        self.index = nxt
        return logits1, logits2, labels
