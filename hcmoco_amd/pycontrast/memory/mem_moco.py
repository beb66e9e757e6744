"""MoCo-style queues (secondary path; reference: /root/reference/pycontrast/memory/mem_moco.py).

``logits = cat(q.k, q @ queue^T)/T`` runs in ``hcm_moco_logits``; its backward wrt ``q`` is one
plain library GEMM (rocBLAS via ``torch.mm``).  The ring pointer is host bookkeeping, bit-exact.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hip_ops


class _MoCoLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, queue, T):
        ctx.save_for_backward(k, queue.clone())
        ctx.T = T
        return hip_ops.moco_logits(q, k, queue, T)

    @staticmethod
    def backward(ctx, g):
        k, queue = ctx.saved_tensors
        gq = (g[:, :1] * k + torch.mm(g[:, 1:], queue)) / ctx.T
        return gq, None, None, None


class BaseMoCo(nn.Module):
    def __init__(self, K=65536, T=0.07):
        super().__init__()
        self.K, self.T, self.index = K, T, 0

    def _compute_logit(self, q, k, queue):
        return _MoCoLogits.apply(q, k.detach(), queue, self.T)

    def _enqueue(self, queues, keys):
        nxt = self.index
        for queue, k in zip(queues, keys):
            nxt = hip_ops.moco_enqueue(queue, k, self.index)
        self.index = nxt


class RGBMoCo(BaseMoCo):
    def __init__(self, n_dim, K=65536, T=0.07):
        super().__init__(K, T)
        self.register_buffer('memory', F.normalize(torch.randn(K, n_dim)))

    def forward(self, q, k, q_jig=None, all_k=None):
        k = k.detach()
        logits = self._compute_logit(q, k, self.memory)
        logits_jig = self._compute_logit(q_jig, k, self.memory) if q_jig is not None else None
        labels = torch.zeros(q.shape[0], dtype=torch.long, device=q.device)
        self._enqueue([self.memory], [all_k if all_k is not None else k])
        return (logits, logits_jig, labels) if q_jig is not None else (logits, labels)


class CMCMoCo(BaseMoCo):
    def __init__(self, n_dim, K=65536, T=0.07):
        super().__init__(K, T)
        self.register_buffer('memory_1', F.normalize(torch.randn(K, n_dim)))
        self.register_buffer('memory_2', F.normalize(torch.randn(K, n_dim)))

    def forward(self, q1, k1, q2, k2, q1_jig=None, q2_jig=None, all_k1=None, all_k2=None):
        k1, k2 = k1.detach(), k2.detach()
        logits1 = self._compute_logit(q1, k2, self.memory_2)
        logits2 = self._compute_logit(q2, k1, self.memory_1)
        jig = q1_jig is not None and q2_jig is not None
        if jig:
            logits1_jig = self._compute_logit(q1_jig, k2, self.memory_2)
            logits2_jig = self._compute_logit(q2_jig, k1, self.memory_1)
        labels = torch.zeros(q1.shape[0], dtype=torch.long, device=q1.device)
        all_k1 = all_k1 if all_k1 is not None else k1
        all_k2 = all_k2 if all_k2 is not None else k2
        assert all_k1.size(0) == all_k2.size(0)
        self._enqueue([self.memory_1, self.memory_2], [all_k1, all_k2])
        if jig:
            return logits1, logits2, logits1_jig, logits2_jig, labels
        return logits1, logits2, labels
