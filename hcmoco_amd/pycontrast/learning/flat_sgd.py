"""One optimizer launch per encoder instead of forty per step.

The reference hands ``torch.optim.SGD(model.parameters(), ...)`` to the trainer (main_contrast.py:66) and calls
``optimizer.step()`` once per iteration (learning/contrast_trainer.py:560-562 / 1028-1030).  On the MI355X the two
HRNets alone are ~1 900 parameter tensors of 41 k floats on average: even the fused multi-tensor SGD needs ~40
launches and 0.92 ms for 78 MB of parameters (0.4 TB/s), and those 0.92 ms sit at the very end of the step where
nothing else runs.

The encoder runtime already keeps every HRNet's parameter GRADIENTS in one dense buffer (``[dw | dgamma | dbeta]``
per layer, program order -- csrc/torch_glue, what learning/grad_sync.py all-reduces in place).  ``FlatParamSGD``
gives the PARAMETERS the same layout: at the first step it moves each such encoder's parameters into one flat
tensor (every ``nn.Parameter`` keeps its identity and becomes a view of it), swaps them for that flat tensor in the
wrapped optimizer's ``param_groups`` and points the flat tensor's ``.grad`` at the flat gradient buffer before
every ``step()``.  The update rule is untouched -- it is the caller's optimizer object that runs, element by
element the same arithmetic, on 2 large tensors + the few small ones outside the encoders.

``state_dict()`` / ``load_state_dict()`` keep the reference's checkpoint layout (one ``momentum_buffer`` per
parameter, indexed in ``model.parameters()`` order), so checkpoints move freely between this build and the
reference.  A parameter group is only flattened when ALL of an encoder's program parameters live in that one
group (same hyper-parameters); anything else stays as it is.
"""
import torch

from . import grad_sync


class FlatParamSGD(object):
    def __init__(self, optimizer, model):
        self.inner = optimizer
        self.model = model
        self._orig = [[p for p in g['params']] for g in optimizer.param_groups]     # reference order, for checkpoints
        self._flat = []            # (encoder, flat parameter tensor, [params], [offsets])
        self._done = False
        self._pending_state = None
        self._grad_ptr = {}        # id(flat parameter) -> storage address of the gradient buffer validated in full

    # ------------------------------------------------------------------ plain delegation
    @property
    def param_groups(self):
        return self.inner.param_groups

    @property
    def state(self):
        return self.inner.state

    def __getattr__(self, name):                       # defaults, add_param_group, ...
        return getattr(self.__dict__['inner'], name)

    def zero_grad(self, set_to_none=True):
        self.inner.zero_grad(set_to_none=set_to_none)
        for _, flat, params, _ in self._flat:          # the originals are no longer in the optimizer's groups
            if set_to_none:
                flat.grad = None
                for p in params:
                    p.grad = None
            else:
                for p in params:
                    if p.grad is not None:
                        p.grad.zero_()

    # ------------------------------------------------------------------ flattening
    def _encoders(self):
        from ..networks.hrnet import HighResolutionNet
        return [m for m in self.model.modules() if isinstance(m, HighResolutionNet) and m.last_program is not None]

    def _flatten(self):
        """Called at the first step(): the encoder programs (and with them the gradient layout) exist now."""
        self._done = True
        for enc in self._encoders():
            params = list(enc.last_program.params)
            ids = set(id(p) for p in params)
            group = None
            for g in self.inner.param_groups:
                have = sum(1 for p in g['params'] if id(p) in ids)
                if have == len(params):
                    group = g
                elif have:
                    group = None
                    break
            if group is None or len(ids) != len(params) or any(p.grad is None for p in params):
                continue
            n = sum(p.numel() for p in params)
            if self._flat_grad(params, n) is None:
                continue
            flat = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
            offs, o = [], 0
            with torch.no_grad():
                for p in params:
                    flat[o:o + p.numel()].copy_(p.reshape(-1))
                    offs.append(o)
                    o += p.numel()
                for p, off in zip(params, offs):
                    p.data = flat[off:off + p.numel()].view(p.shape)
            # momentum buffers that already exist (resumed run) move into one flat buffer as well
            bufs = [self.inner.state.get(p, {}).get('momentum_buffer') for p in params]
            if any(b is not None for b in bufs):
                fb = torch.zeros_like(flat)
                for b, p, off in zip(bufs, params, offs):
                    if b is not None:
                        fb[off:off + p.numel()].copy_(b.reshape(-1))
                self.inner.state[flat] = {'momentum_buffer': fb}
            for p in params:
                self.inner.state.pop(p, None)
            pos = min(i for i, p in enumerate(group['params']) if id(p) in ids)
            group['params'] = [p for p in group['params'] if id(p) not in ids]
            group['params'].insert(pos, flat)
            self._flat.append((enc, flat, params, offs))

    @staticmethod
    def _flat_grad(params, n, full_check=True):
        """The encoder's flat gradient buffer as ONE 1-D tensor, or None when the parameters' gradients are not the
        consecutive pieces of one allocation (they are after a backward of the encoder program: csrc/torch_glue
        hands autograd views of its dense buffer, AccumulateGrad keeps them)."""
        g0 = params[0].grad
        if g0 is None:
            return None
        st = g0.untyped_storage()
        o = g0.storage_offset()
        if st.nbytes() < (o + n) * g0.element_size():
            return None
        if full_check:
            ptr = st.data_ptr()
            for p in params:
                g = p.grad
                if (g is None or g.untyped_storage().data_ptr() != ptr or g.storage_offset() != o
                        or not g.is_contiguous() or g.dtype != g0.dtype):
                    return None
                o += p.numel()
        else:
            last = params[-1].grad
            if (last is None or last.untyped_storage().data_ptr() != st.data_ptr()
                    or last.storage_offset() != g0.storage_offset() + n - params[-1].numel()):
                return None
        return g0.as_strided((n,), (1,), g0.storage_offset())

    # ------------------------------------------------------------------ the step
    def step(self, closure=None):
        if not self._done:
            if self._pending_state is not None:
                self.inner.load_state_dict(self._pending_state)
                self._pending_state = None
            self._flatten()
        for enc, flat, params, offs in self._flat:
            # someone re-allocated the parameters (model.to(...), .float(), load with assign=True): take their current
            # values back into the flat tensor and re-bind, or the update below would go to memory nobody reads
            if (params[0].data_ptr() != flat.data_ptr()
                    or params[-1].data_ptr() != flat.data_ptr() + offs[-1] * flat.element_size()):
                with torch.no_grad():
                    for p, off in zip(params, offs):
                        flat[off:off + p.numel()].copy_(p.reshape(-1))
                        p.data = flat[off:off + p.numel()].view(p.shape)
            # the cheap first/last test is enough while the gradients live in the allocation that was validated
            # piece by piece; anything else (GradSync re-bound them into a bucket in parameters() order after a
            # module-path step, a new program) gets the full walk again -- first and last can line up while the
            # middle is permuted
            g0 = params[0].grad
            # what identifies a validated layout: the allocation (address AND size: the caching allocator re-issues
            # addresses), where the first gradient sits in it, the encoder program that laid it out, and GradSync's
            # bucket generation (a re-built bucket can land on the old address with another interior order -- ADVICE r03)
            ptr = None if g0 is None else (g0.untyped_storage().data_ptr(), g0.untyped_storage().nbytes(),
                                           g0.storage_offset(), id(getattr(enc, 'last_program', None)),
                                           grad_sync.GENERATION[0])
            known = ptr is not None and self._grad_ptr.get(id(flat)) == ptr
            fg = self._flat_grad(params, flat.numel(), full_check=not known)
            if fg is not None:
                self._grad_ptr[id(flat)] = ptr
            if fg is None:
                # this step's forward did not run as an encoder program (module path, another input shape): gather
                # the separate gradients; a parameter without one contributes zeros (it still sees weight decay)
                if all(p.grad is None for p in params):
                    flat.grad = None
                    continue
                fg = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
            flat.grad = fg
        return self.inner.step(closure)

    # ------------------------------------------------------------------ checkpoints in the reference's layout
    def state_dict(self):
        if not self._flat:
            return self.inner.state_dict()
        index, k = {}, 0
        for g in self._orig:
            for p in g:
                index[id(p)] = k
                k += 1
        sd = self.inner.state_dict()
        state = {}
        where = {}
        for _, flat, params, offs in self._flat:
            for p, off in zip(params, offs):
                where[id(p)] = (flat, off)
        for g in self._orig:
            for p in g:
                if id(p) in where:
                    flat, off = where[id(p)]
                    st = self.inner.state.get(flat)
                    if st and st.get('momentum_buffer') is not None:
                        state[index[id(p)]] = {'momentum_buffer': st['momentum_buffer'][off:off + p.numel()].view(p.shape).clone()}
                else:
                    st = self.inner.state.get(p)
                    if st:
                        state[index[id(p)]] = {k2: (v.clone() if torch.is_tensor(v) else v) for k2, v in st.items()}
        groups = []
        k = 0
        for g, orig in zip(sd['param_groups'], self._orig):
            g = dict(g)
            g['params'] = list(range(k, k + len(orig)))
            k += len(orig)
            groups.append(g)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        if not self._done:
            try:
                self.inner.load_state_dict(sd)         # nothing flattened yet: the reference layout IS the inner layout
            except (ValueError, KeyError):             # group sizes do not match yet (e.g. a wrapped optimizer that
                self._pending_state = sd               # is re-grouped before the first step): retry at step()
            return
        if not self._flat:
            self.inner.load_state_dict(sd)
            return
        flat_of = {}
        for _, flat, params, offs in self._flat:
            for p, off in zip(params, offs):
                flat_of[id(p)] = (flat, off)
        k = 0
        for g_sd, g, orig in zip(sd['param_groups'], self.inner.param_groups, self._orig):
            for key, v in g_sd.items():
                if key != 'params':
                    g[key] = v
            for p in orig:
                st = sd['state'].get(k, sd['state'].get(str(k)))
                k += 1
                if not st:
                    continue
                if id(p) in flat_of:
                    flat, off = flat_of[id(p)]
                    fb = self.inner.state.setdefault(flat, {}).get('momentum_buffer')
                    if fb is None:
                        fb = self.inner.state[flat]['momentum_buffer'] = torch.zeros_like(flat)
                    if st.get('momentum_buffer') is not None:
                        fb[off:off + p.numel()].copy_(st['momentum_buffer'].reshape(-1))
                else:
                    self.inner.state[p] = {k2: (v.to(p.device) if torch.is_tensor(v) else v) for k2, v in st.items()}
