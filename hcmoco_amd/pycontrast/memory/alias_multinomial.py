"""Walker alias sampler over the bank rows.

Reference: /root/reference/pycontrast/memory/alias_multinomial.py.  Table construction is the
same algorithm (host side, ``hcm_alias_build``); ``draw`` runs on the GPU in one kernel
(``hcm_alias_draw``) with a counter-based Philox generator instead of five eager ops on torch's
global generator, so a draw is a pure function of ``(seed, offset)`` and reproducible bit for bit
(oracle: ``oracle/hcmoco_oracle.py:alias_draw_philox``).
"""
import torch

from ... import hip_ops


class AliasMethod(object):
    def __init__(self, probs, seed=None):
        self.prob, self.alias = hip_ops.alias_build(probs)
        if seed is None:
            # replicas must draw DIFFERENT negatives (the reference's per-process generators are
            # unsynchronised): fold the rank into the Philox key
            import torch.distributed as dist
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            seed = torch.initial_seed() ^ (rank * 0x9E3779B97F4A7C15)
        self.seed = int(seed) & (2 ** 64 - 1)
        self.offset = 0

    def cuda(self, device=None):
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        return self.to(dev)

    def to(self, device):
        self.prob = self.prob.to(device)
        self.alias = self.alias.to(device)
        return self

    def draw(self, N):
        """N samples (alias_multinomial.py:48-65).  Each call consumes one Philox offset."""
        idx = hip_ops.alias_draw(self.prob, self.alias, None, 1, int(N), self.seed, self.offset)
        self.offset += 1
        return idx.view(-1)

    def draw_with_positive(self, y, K1):
        """idx [B, K1] with idx[:,0] = y (mem_bank.py:176-177) in one launch."""
        idx = hip_ops.alias_draw(self.prob, self.alias, y.contiguous(), y.shape[0], int(K1), self.seed, self.offset)
        self.offset += 1
        return idx
