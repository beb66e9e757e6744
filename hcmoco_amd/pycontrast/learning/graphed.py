"""hipGraph capture of the encoder forward/backward.

One stage-2 step launches ~9400 kernels, most of them a few microseconds long (HRNet is a zoo of
small convolutions, batch-norms and element-wise ops): in eager mode the host, not the MI355X,
paces the step (profiles/r01_bench_one_step_summary.csv: 130 ms of GPU work in a 138 ms step, ~14
us per launch).  ``GraphedEncoder`` captures the encoder's forward and backward as two hipGraphs
(``torch.cuda.make_graphed_callables``) and leaves the loss section -- the hand-written HIP kernels,
~40 launches -- eager between them, so the collectives, the bank update and the per-kernel hipEvent
timing stay ordinary stream work.

Gradient synchronisation for N > 1 is explicit here (``allreduce_grads``): one flat RCCL all-reduce
of the 78 MB of gradients after backward.  At 8 GPUs over xGMI that is ~1-2 ms in a >100 ms step, so
nothing is lost by not overlapping it with backward, and DistributedDataParallel's per-bucket hooks
(which cannot live inside a captured backward) are not needed.
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class _EncoderOutputs(nn.Module):
    """Flat tensor-in / tensor-out view of the model, as graph capture requires."""

    def __init__(self, model, stage2):
        super().__init__()
        self.model, self.stage2 = model, stage2

    def forward(self, x, s):
        if not self.stage2:
            return (self.model(x, s),)
        _f1, _f2, feat3, f, aux = self.model(x, s, return_fm=True)
        return f, aux['linear_merge1'], aux['linear_merge2'], feat3


class GraphedEncoder(object):
    def __init__(self, model, sample_x, sample_s, stage2=True, warmup=3):
        self.stage2 = stage2
        wrapper = _EncoderOutputs(model, stage2)
        self.static_x = sample_x.detach().clone()
        self.static_s = sample_s.detach().clone()
        self.call = torch.cuda.make_graphed_callables(wrapper, (self.static_x, self.static_s),
                                                      num_warmup_iters=warmup)

    def __call__(self, x, s):
        # replays read from the captured addresses: stage the batch into the static inputs
        self.static_x.copy_(x, non_blocking=True)
        self.static_s.copy_(s, non_blocking=True)
        return self.call(self.static_x, self.static_s)


def broadcast_model(model, src=0):
    """Replica consistency at start-up when DistributedDataParallel is not wrapping the model."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src)


def allreduce_grads(params, world):
    """Mean of the gradients over ranks with ONE collective (flat fp32 buffer)."""
    if world <= 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat.div_(world)
    torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
