"""Gradient averaging across replicas, overlapped with the backward pass.

The reference wraps the model in ``DistributedDataParallel`` (learning/contrast_trainer.py:74):
gradients are all-reduced bucket by bucket while autograd is still walking backwards.  DDP's hooks
fire on the autograd thread when a parameter's ``.grad`` is *assigned*; in this build the HRNet
encoders' gradients are *written* later, by the encoder runtime's helper threads (csrc/torch_glue:
one C++ reverse loop per encoder, on that encoder's HIP stream), so DDP would reduce buffers that are
not filled yet.  ``GradSync`` keeps the overlap and drops the hooks:

* **Encoder buckets** -- every HRNet that ran as an encoder program owns ONE dense flat gradient
  buffer (``[dw | dgamma | dbeta]`` per layer, program order).  Its reverse loop fills that buffer
  from the back and records a HIP event each time another n-th is complete
  (``torch.ops.hcmoco.set_grad_chunks``).  After ``loss.backward()`` returned -- the helper threads
  are typically still issuing -- this class walks the chunks in completion order:
  ``grad_chunk_wait`` blocks until chunk k is *issued*, makes a dedicated comm-launch stream wait for
  its event, and the chunk is all-reduced **in place** (RCCL, ``async_op=True``: the process group's
  own communication stream runs it as soon as the event fires).  Encoder1's last layers are on the
  wire while both encoders' earlier layers are still being differentiated; no concatenation, no copy
  back -- the parameters' ``.grad`` are views of the buffer that was reduced.
* **Rest bucket** -- SemGCN, heads, 1x1 projections (and anything that did not run as a program,
  which is everything on CPU): copied into one persistent flat buffer with one multi-tensor launch,
  reduced, and ``.grad`` re-bound to views of it.  Built over a FIXED parameter list; a parameter that
  got no gradient on this rank contributes zeros and receives the average, so every replica applies
  the same update whatever its local graph looked like; one that got none on ANY rank keeps
  ``.grad = None`` (agreed once, see ``_present``), so N > 1 applies the update N = 1 applies.
* ``optimizer.step()`` is ordered behind all of it by ``work.wait()`` on the caller's stream.

``mode='flat'`` keeps the single-bucket behaviour (one all-reduce after the join) for comparison;
both modes produce bit-identical parameters with two ranks and parameters equal to round-off with eight (a ring adds the
eight contributions of an element in an order that depends on the element's position in the buffer, and the modes cut the
buffers differently; tests/test_plumbing_cpu.py, world_size 2 and 8 over gloo).

Not synchronised, unlike DDP's default ``broadcast_buffers=True``: BatchNorm running statistics stay
per-replica.  They never enter a training-mode forward, and rank 0 writes the checkpoint from its
own buffers in both implementations, so checkpoints agree with the reference's.
"""
import os
import torch
import torch.distributed as dist


# torch's grouped-launch context (ncclGroupStart/End) is a private symbol: bind it once, fall back to one launch per piece
_coalescing_manager = getattr(dist, '_coalescing_manager', None)

GENERATION = [0]      # bumped whenever a flat bucket is (re-)allocated: consumers that cached a validated layout re-check


def broadcast_model(model, src=0):
    """Replica consistency at start-up when DistributedDataParallel is not wrapping the model."""
    if not dist.is_initialized():
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src)


class GradSync(object):
    def __init__(self, model, params, mode='overlap', chunks=4, glue=None):
        """model: the (unwrapped) network; params: the optimizer's parameters, in its order;
        glue: ``torch.ops.hcmoco`` on ROCm (None on CPU)."""
        assert mode in ('overlap', 'flat')
        self.world = dist.get_world_size()
        self.mode = mode
        self.params = [p for p in params if p.requires_grad]
        self.glue = glue
        self.chunks = int(chunks)
        self.avg = dist.get_backend() == 'nccl'          # RCCL averages in the collective; gloo sums
        self.encoders = []
        if glue is not None and mode == 'overlap' and self.chunks > 0:
            from ..networks.hrnet import HighResolutionNet
            self.encoders = [m for m in model.modules() if isinstance(m, HighResolutionNet)]
            glue.set_grad_chunks(self.chunks)
        # CPU / module-path buckets of the overlapped schedule: one per top-level child of the model
        # (encoder1, encoder2, ...), launched in reverse registration order like DDP's buckets
        self.groups = None
        if mode == 'overlap':
            seen, groups = set(), []
            for child in reversed(list(model.children())):
                g = [p for p in child.parameters() if p.requires_grad and id(p) not in seen]
                seen.update(id(p) for p in g)
                if g:
                    groups.append(g)
            known = set(id(p) for p in self.params)
            stray = [p for p in self.params if id(p) not in seen]
            groups = [[p for p in g if id(p) in known] for g in groups]
            if stray:
                groups.append(stray)
            self.groups = [g for g in groups if g]
        self._flat = {}
        self._presence = {}          # bucket key -> agreed pattern (union over the ranks): see _present
        self._changed = False        # this rank saw a gradient outside an agreed pattern during this reduce()
        self._flag_host = self._flag_event = None
        self._flag_pinned = self._flag_done = None
        self.agreements = 0          # presence all-reduces issued so far (tests read it)
        # bench.py sets this to a list: every reduce() then appends a HIP event pair around the waits below -- the time the
        # trainer's stream stands still for collectives that backward did not cover (exposed communication)
        self.wait_events = None
        # bench.py (N > 1): (end of the previous step's collectives, this step's LAST encoder chunk ready) event pairs -- how long
        # into a step this rank's gradients are complete; max - min over the ranks is what the fast ranks wait inside RCCL
        self.ready_events = None
        self._last_end = None
        self._comm = None
        # several chunks in one RCCL launch (nccl only: gloo's coalescing manager has no all-reduce fast path)
        self.coalesce = self.avg and os.environ.get('HCM_GRAD_COALESCE', '1') != '0'
        self.launched = 0            # collectives launched by the last reduce() (tests / bench read it)

    # ------------------------------------------------------------------ helpers
    def _launch(self, t):
        self.launched += 1
        if self.avg:
            return [t], dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True)
        return [t], dist.all_reduce(t, async_op=True)

    def _launch_group(self, pieces):
        """The k-th chunks of all encoders as ONE RCCL launch (ncclGroupStart ... ncclGroupEnd through torch's
        coalescing manager): the two encoders walk their reverse loops at the same pace, so their k-th chunks are ready
        together, and a collective costs ~0.15 ms of launch and stream hand-over whatever its size."""
        if len(pieces) == 1 or not self.coalesce:
            return [self._launch(t) for t in pieces]
        if _coalescing_manager is None:          # private torch API: without it, one launch per piece
            return [self._launch(t) for t in pieces]
        self.launched += 1
        with _coalescing_manager(device=pieces[0].device, async_ops=True) as cm:
            for t in pieces:
                if self.avg:
                    dist.all_reduce(t, op=dist.ReduceOp.AVG)
                else:
                    dist.all_reduce(t)
        return [(list(pieces), cm)]

    def _present(self, key, group):
        """Which parameters of ``group`` received a gradient on SOME rank.  A parameter that no rank used (stage 1
        with ``--linear_feat_map 1``: the two 1x1 projections) must keep ``.grad = None`` -- SGD then skips it, as it
        does with one GPU and as DistributedDataParallel does for globally unused parameters; averaging zeros into it
        would hand it weight decay and momentum that the single-GPU run never applies.

        Every collective here is entered by ALL ranks or by none (the r03 version re-agreed whenever the
        LOCAL pattern changed, i.e. one rank alone could issue the MAX all-reduce while its peers were already in the
        bucket average).  The agreed pattern is the union over the ranks and is decided (a) at the first step, when
        every rank's cache is empty, and (b) again when the ``changed`` flag that every step's last bucket carries
        came back non-zero -- a reduced value, hence the same decision on every rank.  In between, a rank whose local
        pattern shrinks contributes zeros (no collective needed), and a rank that suddenly has a gradient OUTSIDE the
        union raises the flag: that one gradient is dropped on that step (``.grad = None``, so no replica applies a
        local-only update) and the union grows on the next.  Consequence, stated so nobody has to find it: at an
        in-process switch that makes a parameter USED for the first time on every rank at once (stage 1 -> stage 2: the two
        1x1 projections), every rank drops that parameter's first gradient, so an N-GPU run differs from the 1-GPU run by
        that one update of those parameters; the released recipe switches stages between processes (``--pretrain``), where
        the first step agrees on the full union."""
        local = tuple(p.grad is not None for p in group)
        cached = self._presence.get(key)
        if cached is not None and len(cached) != len(group):
            # the composition of a group is a property of the graph, identical on every rank: a different length is a
            # cache miss on all of them alike (zip() would silently truncate and never reduce the tail)
            cached = None
        if cached is None:
            flags = torch.tensor([1.0 if v else 0.0 for v in local], dtype=torch.float32, device=group[0].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            self.agreements += 1
            cached = self._presence[key] = tuple(bool(v) for v in (flags > 0).tolist())
        for p, mine, ok in zip(group, local, cached):
            if mine and not ok:
                self._changed = True
                p.grad = None
        return cached

    def _poll_changed(self):
        """The ``changed`` flag of the PREVIOUS step (reduced with that step's last bucket, copied to the host without
        blocking): non-zero on every rank or on none.  Non-zero -> forget the agreed patterns, so that this step's
        ``_present`` calls re-agree, on every rank alike."""
        if self._flag_event is not None:
            # the copy was queued a whole step ago: normally done already (query() is a host-side check, no stall)
            if not self._flag_event.query():
                self._flag_event.synchronize()
            self._flag_event = None
        if self._flag_host is not None and float(self._flag_host[0]) > 0:
            self._presence.clear()
        self._flag_host = None

    def _bucket(self, key, group):
        """Copy the group's gradients into its persistent flat buffer and re-bind ``.grad`` to views of it; returns
        the buffer (None when nothing in the group has a gradient anywhere).  Parameters without a gradient on ANY
        rank are left out and keep ``.grad = None``; one that only THIS rank did not use contributes zeros and
        receives the average."""
        present = self._present(key, group)
        group = [p for p, ok in zip(group, present) if ok]
        if not group:
            return None
        key = (key, present)
        n = sum(p.numel() for p in group)
        flat = self._flat.get(key)
        if flat is None or flat.numel() != n + 1 or flat.device != group[0].device:
            # one slot more than the gradients: the step's ``changed`` flag rides in the last bucket (see _present)
            flat = self._flat[key] = torch.zeros(n + 1, dtype=group[0].dtype, device=group[0].device)
            GENERATION[0] += 1
        views = [v.view_as(p) for v, p in zip(flat[:n].split([p.numel() for p in group]), group)]
        have = [(v, p.grad) for v, p in zip(views, group) if p.grad is not None]
        if len(have) != len(group):
            flat[:n].zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for p, v in zip(group, views):
            p.grad = v
        return flat

    # ------------------------------------------------------------------ the step
    def reduce(self, join=None):
        """Call right after ``loss.backward()``.  ``join``: callable that blocks until every deferred
        gradient is in its stream and orders the current stream behind them (``wgrad_join``)."""
        self.launched = 0
        self._poll_changed()
        self._changed = False
        works, handled = [], set()
        if self.encoders:
            dev = self.params[0].device
            if self._comm is None:
                # The event waits in front of the collectives go onto the trainer's own stream: it has nothing else to do
                # until the join, and RCCL's stream picks the chunks up from there.  r02 gave them a stream of their
                # own: a fifth active stream on four hardware queues made two of them share a queue, and that alone was
                # most of the 1-rank cost of the N>1 path (665 vs 689 of 691 samples/s, r03).
                self._comm = torch.cuda.current_stream(dev)
            live = []
            for enc in self.encoders:
                n = int(self.glue.grad_chunk_count(enc.grad_tag)) if enc.last_program is not None else 0
                if n:
                    live.append((enc, n))
            # chunk-major: chunk k of every encoder before chunk k+1 of any (= completion order)
            for k in range(max([n for _, n in live] or [0])):
                with torch.cuda.stream(self._comm):
                    pieces = [self.glue.grad_chunk_wait(enc.grad_tag, k) for enc, n in live if k < n]
                    works.extend(self._launch_group(pieces))
            for enc, _ in live:
                handled.update(id(p) for p in enc.last_program.params)
            if self.ready_events is not None and live:
                ready = torch.cuda.Event(enable_timing=True)
                ready.record(self._comm)          # behind the stream-wait on the last chunk's event
                if self._last_end is not None:
                    self.ready_events.append((self._last_end, ready))
        if join is not None:
            join()
        if self.mode == 'flat':
            groups = {'flat': self.params}
        elif handled:       # the encoders went chunk by chunk: everything else is small -> ONE more collective
            groups = {'rest': [p for p in self.params if id(p) not in handled]}
        else:
            groups = dict(enumerate(self.groups))
        flats = []
        for key, g in groups.items():
            if g:
                flat = self._bucket(key, g)
                if flat is not None:
                    flats.append(flat)
        if not flats:            # nothing has a gradient anywhere: the flag still has to travel
            flat = self._flat.get('flag')
            if flat is None:
                flat = self._flat['flag'] = torch.zeros(1, dtype=torch.float32, device=self.params[0].device)
            flats.append(flat)
        carrier = flats[-1]
        carrier[-1:].fill_(1.0 if self._changed else 0.0)     # MAX-like under AVG / SUM: non-zero iff some rank raised it
        for flat in flats:
            works.append(self._launch(flat))
        timed = self.wait_events is not None and carrier.is_cuda
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for ts, w in works:
            w.wait()                 # RCCL: orders the current stream behind the collective, no host block
            if not self.avg:
                for t in ts:
                    t.div_(self.world)
        if timed:
            ev[1].record()
            self.wait_events.append(ev)
            self._last_end = ev[1]
        # the reduced flag goes to the host without a sync; it is read at the top of the next reduce()
        if carrier.is_cuda:
            if self._flag_pinned is None:            # one pinned word for the life of the object (a pinned allocation per
                self._flag_pinned = torch.empty(1, dtype=carrier.dtype, pin_memory=True)      # step costs more than the step's
                self._flag_done = torch.cuda.Event()                                            # whole exposed communication)
            self._flag_host = self._flag_pinned
            self._flag_host.copy_(carrier[-1:], non_blocking=True)
            self._flag_event = self._flag_done
            self._flag_event.record()
        else:
            self._flag_host = carrier[-1:].clone()
        return self.launched
