"""Three-modality instance-discrimination memory bank.

Reference: /root/reference/pycontrast/memory/mem_bank.py:157-205 (``CMCMem3``), :15-40
(``BaseMem``).  Same constructor, buffers (``memory_1/2/3`` [n_data, n_dim], L2-normalised randn),
``forward`` signature and return tuple.  Two ways in:

``forward(...)``        the literal reference contract: six materialised logit tensors + labels,
                        differentiable through ``x1..x3`` (HIP kernels ``hcm_bank_logits_fwd/bwd``).
``forward_loss(...)``   what this build's trainer uses: the fused kernel computes the six
                        cross-entropy losses, accuracies and d/dx in one pass over the gathered
                        rows and never materialises a logit (``hcm_bank_nce_fused``).

Both draw the K negatives with ``AliasMethod`` on the device, place the positive in column 0 and
apply the momentum update AFTER the reads (mem_bank.py:195-203).  ``idx=`` injects the row
indices explicitly (parity mode).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hip_ops
from .alias_multinomial import AliasMethod


class BaseMem(nn.Module):
    def __init__(self, K=65536, T=0.07, m=0.5):
        super().__init__()
        self.K, self.T, self.m = K, T, m


class CMCMem3(BaseMem):
    def __init__(self, n_dim, n_data, K=65536, T=0.07, m=0.5, seed=None, bank_dtype=torch.float32):
        """``bank_dtype=torch.bfloat16`` (BASELINE config 5, a build-side option) stores the banks in
        bf16 -- half the gather traffic; all products and sums stay fp32 in the kernels."""
        super().__init__(K, T, m)
        self.n_dim, self.n_data = n_dim, n_data
        self.multinomial = AliasMethod(torch.ones(n_data), seed=seed)
        for name in ('memory_1', 'memory_2', 'memory_3'):
            self.register_buffer(name, F.normalize(torch.randn(n_data, n_dim)).to(bank_dtype))
        self._oob = None             # device flag: some index had to be clamped (see _in_range)
        self._oob_flag = None        # the same, set by the range-checked kernels (draw / update_strided)
        self._pixel_draws = 0

    # nn.Module.cuda()/.to() move the buffers; the sampler tables follow
    def _apply(self, fn):
        super()._apply(fn)
        self.multinomial.to(self.memory_1.device)
        return self

    def banks(self):
        return [self.memory_1, self.memory_2, self.memory_3]

    # The reference's index_select / index_copy_ device-assert on a row index outside [0, n_data); the
    # kernels here take raw pointers, so the module range-checks what it hands them: indices are clamped
    # (nothing outside the banks is ever read or written) and a sticky device flag records that a clamp
    # changed something.  ``check_indices()`` reads the flag (one host sync: the trainer calls it where
    # it syncs anyway) and raises like the reference would have.
    def _in_range(self, t):
        if t is None:
            return None
        safe = t.clamp(0, self.n_data - 1)
        bad = (safe != t).any()
        self._oob = bad if self._oob is None else (self._oob | bad)
        return safe

    def check_indices(self):
        """Raise if any index handed to the kernels since the last call had to be clamped.  With several ranks the
        flag is MAX-reduced first, so that EVERY rank raises (one rank raising alone would leave the others hung in
        their next collective); every rank must therefore call this at the same points of its loop -- the trainer
        does, at ``print_freq`` and at the end of an epoch."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if self._oob is None and self._oob_flag is None and not multi:
            return
        flag = self._oob if self._oob is not None else torch.zeros((), dtype=torch.bool, device=self.memory_1.device)
        self._oob = None
        if self._oob_flag is not None:
            flag = flag | (self._oob_flag[0] != 0)
            self._oob_flag = None
        if multi:
            f = flag.to(torch.int32).reshape(1)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            flag = f[0] > 0
        if bool(flag):
            raise IndexError('CMCMem3: a bank row index was outside [0, %d) (dataset index / injected idx)%s'
                             % (self.n_data, ' on some rank' if multi else ''))

    def _flag(self):
        """Sticky int32 device flag the checked kernels OR a 1 into when they had to clamp a row index."""
        if self._oob_flag is None or self._oob_flag.device != self.memory_1.device:
            self._oob_flag = torch.zeros(1, dtype=torch.int32, device=self.memory_1.device)
        return self._oob_flag

    def draw(self, y):
        """idx [B, K+1] with idx[:,0] = y clamped into the bank (mem_bank.py:176-177) -- one launch."""
        mn = self.multinomial
        idx = hip_ops.alias_draw(mn.prob, mn.alias, y.contiguous(), y.shape[0], self.K + 1, mn.seed, mn.offset,
                                 oob=self._flag())
        mn.offset += 1
        return idx

    def update_strided(self, all_xs, ldx, all_y):
        """Momentum update from three [BW, D] column blocks with row stride ``ldx`` (slices of the gathered
        feature matrix), indices range-checked inside the kernel."""
        hip_ops.bank_update(self.banks(), all_xs, all_y, self.m, ldx=ldx, oob=self._flag())

    def next_pixel_key(self):
        """(seed, offset) of the next pixel draw: the sampler's Philox key, its own offset stream (high bit set,
        so it never collides with the negative draws of the same step)."""
        mn = self.multinomial
        self._pixel_draws += 1
        return mn.seed, (1 << 63) | self._pixel_draws

    def _indices(self, y, idx):
        if idx is not None:
            assert idx.shape == (y.shape[0], self.K + 1)
            return self._in_range(idx).contiguous()
        return self.multinomial.draw_with_positive(y, self.K + 1)

    def _update(self, x1, x2, x3, y, all_x1, all_x2, all_x3, all_y):
        if all_x1 is not None and all_x2 is not None and all_x3 is not None and all_y is not None:
            hip_ops.bank_update(self.banks(), [all_x1, all_x2, all_x3], all_y, self.m)
        else:
            hip_ops.bank_update(self.banks(), [x1, x2, x3], y, self.m)

    def forward(self, x1, x2, x3, y, all_x1=None, all_x2=None, all_x3=None, all_y=None, idx=None):
        same = all_y is y
        y = self._in_range(y)
        all_y = y if same else self._in_range(all_y)
        idx = self._indices(y, idx)
        logits = hip_ops.bank_logits([x1, x2, x3], self.banks(), idx, self.T)
        labels = torch.zeros(x1.shape[0], dtype=torch.long, device=x1.device)
        self._update(x1, x2, x3, y, all_x1, all_x2, all_x3, all_y)
        return logits[0], logits[1], logits[2], logits[3], logits[4], logits[5], labels

    def forward_loss(self, x1, x2, x3, y, all_x1=None, all_x2=None, all_x3=None, all_y=None,
                     use_depth=None, use_rgb=None, idx=None):
        """-> (total, losses[6], accs[6]); total = sum(losses) is differentiable in x1..x3."""
        same = all_y is y
        y = self._in_range(y)
        all_y = y if same else self._in_range(all_y)
        idx = self._indices(y, idx)
        total, losses, accs = hip_ops.bank_nce_fused([x1, x2, x3], self.banks(), idx, self.T, use_depth, use_rgb)
        self._update(x1, x2, x3, y, all_x1, all_x2, all_x3, all_y)
        return total, losses, accs
