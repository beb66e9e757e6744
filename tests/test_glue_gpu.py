"""torch.ops.hcmoco.conv2d (MIOpen issued from the C++ glue with cached plans) against ATen's conv2d.

Floating point: both sides run MIOpen fp32 kernels but may pick different algorithms (Winograd vs
implicit GEMM), so the bound is 1e-3 of the tensor's scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # N, C, H, W, K, R, stride, pad
    (32, 18, 64, 64, 18, 3, 1, 1), (8, 36, 32, 32, 36, 3, 1, 1), (4, 144, 8, 8, 144, 3, 1, 1),
    (8, 18, 64, 64, 36, 3, 2, 1), (8, 64, 32, 32, 256, 1, 1, 0), (4, 3, 64, 64, 64, 3, 2, 1),
    (2, 72, 16, 16, 18, 1, 1, 0), (3, 5, 9, 11, 7, 3, 1, 1),
]


def _grads_agree(ga, gb, min_cos=0.995):
    """Two fp32 implementations of a ~150-layer network drift apart element-wise (ReLU masks flip, the
    coarsest branch normalises over a few hundred values); a wrong formula shows up as a gradient
    pointing elsewhere, which the per-parameter cosine catches."""
    gscale = max(v.norm().item() for v in gb.values())
    worst = min((F.cosine_similarity(ga[n].flatten(), g.flatten(), dim=0).item(), n)
                for n, g in gb.items() if g.norm().item() > 1e-4 * gscale)
    assert worst[0] >= min_cos, worst


def _close(a, b, tol=1e-3):
    scale = b.abs().max().item() + 1e-12
    assert (a - b).abs().max().item() <= tol * scale, ((a - b).abs().max().item(), scale)


@pytest.mark.parametrize('case', CASES)
def test_conv2d_matches_aten(case):
    from hcmoco_amd import _lib
    ops = _lib.torch_glue()
    N, C, H, W, K, R, stride, pad = case
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    w = (torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5).to(dev)
    for rep in range(2):              # second pass runs from the cached plan
        xa, wa = x.clone().requires_grad_(), w.clone().requires_grad_()
        xb, wb = x.clone().requires_grad_(), w.clone().requires_grad_()
        ya = ops.conv2d(xa, wa, stride, pad)
        yb = F.conv2d(xb, wb, None, stride, pad)
        gy = torch.randn(yb.shape, generator=g).to(dev)
        ya.backward(gy)
        yb.backward(gy)
        _close(ya, yb)
        _close(xa.grad, xb.grad)
        _close(wa.grad, wb.grad)


def test_conv2d_skips_unneeded_gradients_and_rejects_cpu():
    from hcmoco_amd import _lib
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    x = torch.randn(2, 4, 8, 8, device=dev)                      # no grad for the input (first layer)
    w = torch.randn(6, 4, 3, 3, device=dev, requires_grad=True)
    y = ops.conv2d(x, w, 1, 1)
    y.sum().backward()
    ref = torch.autograd.grad(F.conv2d(x, w, None, 1, 1).sum(), w)[0]
    _close(w.grad, ref)
    with pytest.raises(RuntimeError):
        ops.conv2d(x.cpu(), w.detach().cpu(), 1, 1)


def test_hrnet_glue_conv_matches_aten():
    from hcmoco_amd.pycontrast.networks import hrnet
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = hrnet.get_hrnet_w18_backbone().to(dev).train()
    x = torch.randn(8, 3, 128, 128, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    res = {}
    for glue in (True, False):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        hrnet.CONV_GLUE = glue
        try:
            ys = net(x)
            sum(y.square().mean() for y in ys).backward()
        finally:
            hrnet.CONV_GLUE = True
        res[glue] = ([y.detach().clone() for y in ys], {n: p.grad.clone() for n, p in net.named_parameters()})
    for a, b in zip(res[True][0], res[False][0]):
        _close(a, b, 1e-2)
    _grads_agree(res[True][1], res[False][1])


@pytest.mark.parametrize('relu', [False, True])
@pytest.mark.parametrize('with_res', [False, True])
def test_conv_bn_act_node_equals_its_two_parts(relu, with_res):
    """The single node issues exactly the launches of conv2d followed by bn_act: forward results are
    bit-identical, gradients agree to rounding (MIOpen's weight-gradient kernel reduces with atomics)."""
    from hcmoco_amd import _lib
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 18, 32, 32, generator=g).to(dev)
    w = (torch.randn(36, 18, 3, 3, generator=g) / 13.0).to(dev)
    res = torch.randn(8, 36, 16, 16, generator=g).to(dev) if with_res else None
    gamma, beta = (torch.rand(36, generator=g) + 0.5).to(dev), torch.randn(36, generator=g).to(dev)
    gy = torch.randn(8, 36, 16, 16, generator=g).to(dev)
    outs = []
    for fused in (True, False):
        xs, ws, gs, bs = (t.clone().requires_grad_() for t in (x, w, gamma, beta))
        rs = res.clone().requires_grad_() if with_res else None
        rm, rv = torch.zeros(36, device=dev), torch.ones(36, device=dev)
        if fused:
            y = ops.conv_bn_act(xs, ws, 2, 1, rs, gs, bs, rm, rv, 0.01, 1e-5, relu)
        else:
            y = ops.bn_act(ops.conv2d(xs, ws, 2, 1), rs, gs, bs, rm, rv, 0.01, 1e-5, relu)
        y.backward(gy)
        outs.append([y.detach(), rm, rv, xs.grad, ws.grad, gs.grad, bs.grad] + ([rs.grad] if with_res else []))
    for i, (a, b) in enumerate(zip(*outs)):
        if i < 3:
            assert torch.equal(a, b)
        else:
            _close(a, b, 1e-5)


def test_upsample_node_matches_interpolate():
    from hcmoco_amd import _lib
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    x = torch.randn(4, 36, 16, 16, device=dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya = ops.upsample_bilinear(xa, 64, 64)
    yb = F.interpolate(xb, size=(64, 64), mode='bilinear', align_corners=False)
    gy = torch.randn_like(yb)
    ya.backward(gy)
    yb.backward(gy)
    _close(ya, yb, 1e-6)
    _close(xa.grad, xb.grad, 1e-6)


def test_deferred_weight_gradients_equal_inline_ones():
    """set_async_wgrad(True): dW calls are issued by the helper thread on the stream of their node; after
    wgrad_join() every gradient equals the inline run (to the rounding of MIOpen's atomic reduction).
    A well-conditioned 8-layer chain on two streams for the element-wise check (a whole HRNet is
    chaotic run to run, see _grads_agree), then the HRNet by gradient direction."""
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.networks import hrnet
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    x = torch.randn(16, 18, 32, 32, generator=g).to(dev)
    ws = [(torch.randn(18, 18, 3, 3, generator=g) / 12.7).to(dev).requires_grad_() for _ in range(16)]
    gam = [(torch.rand(18, generator=g) + 0.5).to(dev).requires_grad_() for _ in range(16)]
    bet = [torch.randn(18, generator=g).to(dev).requires_grad_() for _ in range(16)]
    side = torch.cuda.Stream()

    def chain(lo, hi):
        y = x
        for i in range(lo, hi):
            y = ops.conv_bn_act(y, ws[i], 1, 1, y if i % 2 else None, gam[i], bet[i], None, None, 0.1, 1e-5, True)
        return y

    runs = []
    for deferred in (False, True, True):
        for t in ws + gam + bet:
            t.grad = None
        ops.set_async_wgrad(deferred)
        try:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                b = chain(8, 16)
            a = chain(0, 8)
            torch.cuda.current_stream().wait_stream(side)
            (a.square().mean() + b.square().mean()).backward()
            ops.wgrad_join()
        finally:
            ops.set_async_wgrad(False)
        runs.append([t.grad.clone() for t in ws + gam + bet])
    for other in runs[1:]:
        for a, b in zip(other, runs[0]):
            _close(a, b, 1e-4)

    torch.manual_seed(0)
    net = hrnet.get_hrnet_w18_backbone().to(dev).train()
    xi = torch.randn(8, 3, 128, 128, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    grads = []
    for deferred in (False, True):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        ops.set_async_wgrad(deferred)
        try:
            ys = net(xi)
            sum(y.square().mean() for y in ys).backward()
            ops.wgrad_join()
        finally:
            ops.set_async_wgrad(False)
        grads.append({n: p.grad.clone() for n, p in net.named_parameters()})
    _grads_agree(grads[1], grads[0])


def _hrnet_run(net, x, state, program, deferred=False, use_async=False, branch_streams=False, wgrad_stream=0):
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.networks import hrnet
    ops = _lib.torch_glue()
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    hrnet.ENCODER_PROGRAM = program
    hrnet.BRANCH_STREAMS = branch_streams       # branch i of every HighResolutionModule on stream i
    net._programs.clear()
    ops.set_async_wgrad(deferred)
    ops.set_wgrad_stream(wgrad_stream > 0, max(wgrad_stream, 1))   # weight gradients on a side stream, n per hand-over
    try:
        if use_async:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                h = net.forward_async(x)
            assert h is not None
            ys = net.forward_wait(h)
            torch.cuda.current_stream().wait_stream(side)
        else:
            ys = net(x)
        sum(y.square().mean() for y in ys).backward()
        ops.wgrad_join()
    finally:
        ops.set_async_wgrad(False)
        ops.set_wgrad_stream(False, 16)
        hrnet.ENCODER_PROGRAM = True
        hrnet.BRANCH_STREAMS = False
        net._programs.clear()
    torch.cuda.synchronize()
    return ([y.detach().clone() for y in ys], {n: p.grad.clone() for n, p in net.named_parameters()},
            {n: b.clone() for n, b in net.named_buffers()})


def test_encoder_program_equals_module_path():
    """run_encoder (one node, C++ forward loop, own reverse loop) against the module-by-module path:
    the same launches in the same order.  (Not bit-identical: MIOpen's forward convolutions are not
    run-to-run deterministic on this part -- the module path differs from ITSELF by ~1.5e-5 of the
    output scale.)  On a shallow, well-conditioned HRNet outputs and running statistics agree element-wise
    and every gradient tensor has cosine >= 0.999 with its reference; on the full one outputs element-wise and
    gradients by direction (>= 0.995).  The deferred
    reverse loop on the helper thread and the asynchronous forward give the same numbers."""
    from hcmoco_amd.pycontrast.networks import hrnet
    dev = torch.device('cuda:0')
    saved = {k: dict(v) for k, v in hrnet.STAGES.items()}
    try:
        for k in ('stage2', 'stage3', 'stage4'):
            hrnet.STAGES[k].update(modules=1, blocks=1)
        hrnet.STAGES['stage1'].update(blocks=1)
        torch.manual_seed(1)
        small = hrnet.HighResolutionNet(18).to(dev).train()
    finally:
        hrnet.STAGES.update(saved)
    for m in small.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
    x = torch.randn(16, 3, 128, 128, device=dev)
    state = {k: v.clone() for k, v in small.state_dict().items()}
    ref = _hrnet_run(small, x, state, program=False)
    for kwargs in (dict(), dict(deferred=True), dict(deferred=True, use_async=True),
                   dict(deferred=True, use_async=True, branch_streams=True), dict(deferred=True, use_async=True, wgrad_stream=4)):
        got = _hrnet_run(small, x, state, program=True, **kwargs)
        for a, b in zip(got[0], ref[0]):
            _close(a, b, 1e-4)
        for n, b in ref[2].items():
            assert torch.allclose(got[2][n].float(), b.float(), rtol=1e-4, atol=1e-6), n
        _grads_agree(got[1], ref[1], min_cos=0.999)

    torch.manual_seed(0)
    net = hrnet.get_hrnet_w18_backbone().to(dev).train()
    xi = torch.randn(8, 3, 128, 128, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ref = _hrnet_run(net, xi, state, program=False)
    got = _hrnet_run(net, xi, state, program=True, deferred=True, use_async=True)
    for a, b in zip(got[0], ref[0]):
        _close(a, b, 1e-3)
    _grads_agree(got[1], ref[1])
    # the trainer's default since r05: the weight gradients on a side stream, 8 (and 3: ragged batches) layers per
    # hand-over -- the same kernels on the same operands; every run re-builds the program (MIOpen's Find may pick other
    # algorithms, a few ReLU masks flip), so the two agree like two runs of one mode do: cosine 0.999 per parameter
    for n in (8, 3):
        side = _hrnet_run(net, xi, state, program=True, deferred=True, use_async=True, wgrad_stream=n)
        _grads_agree(side[1], ref[1])
        _grads_agree(side[1], got[1], min_cos=0.999)
    # an input size whose coarsest maps are 7x7 (H*W % 4 != 0) must take the module path, not fail
    y = net(torch.randn(2, 3, 224, 224, device=dev))
    assert [t.shape[-1] for t in y] == [56, 28, 14, 7]


def test_encoder_program_partial_outputs_and_second_backward():
    """A caller that consumes only ONE of the four maps: layers whose value reaches no used output are
    skipped by the reverse loop, and their gradients must be ZERO (not uninitialised memory) and equal
    the module path's.  A second backward through the same forward raises a clear error."""
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.networks import hrnet
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    saved = {k: dict(v) for k, v in hrnet.STAGES.items()}
    try:
        for k in ('stage2', 'stage3', 'stage4'):
            hrnet.STAGES[k].update(modules=1, blocks=1)
        hrnet.STAGES['stage1'].update(blocks=1)
        torch.manual_seed(2)
        net = hrnet.HighResolutionNet(18).to(dev).train()
    finally:
        hrnet.STAGES.update(saved)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
    x = torch.randn(8, 3, 64, 64, device=dev)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    # poison the caching allocator's free blocks so that "uninitialised" would not happen to be zero
    junk = torch.full((64 << 20,), float('nan'), device=dev)
    del junk
    grads = []
    for program in (False, True):
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        hrnet.ENCODER_PROGRAM = program
        net._programs.clear()
        try:
            ys = net(x)
            loss = ys[0].square().mean()
            loss.backward()
            ops.wgrad_join()
            if program:
                with pytest.raises(RuntimeError, match='second backward'):
                    ys[0].square().mean().backward()
        finally:
            hrnet.ENCODER_PROGRAM = True
            net._programs.clear()
        torch.cuda.synchronize()
        grads.append({n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()})
    ref, got = grads
    unused = [n for n, g in ref.items() if g is None or float(g.abs().max()) == 0.0]
    assert unused, 'the test needs parameters that do not reach maps[0]'
    for n in unused:
        assert got[n] is not None and bool(torch.isfinite(got[n]).all()) and float(got[n].abs().max()) == 0.0, n
    used = {n: g for n, g in ref.items() if n not in unused}
    _grads_agree({n: got[n] for n in used}, used, min_cos=0.999)


def test_gradient_chunks_cover_the_flat_buffer_in_completion_order():
    """set_grad_chunks(n): the reverse loop publishes the encoder's dense flat gradient buffer in n pieces,
    last layers first; grad_chunk_wait hands each out as a view once it is issued.  The pieces tile the
    buffer exactly, and they are the SAME memory the parameters' .grad live in."""
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.networks import hrnet
    ops = _lib.torch_glue()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = hrnet.get_hrnet_w18_backbone().to(dev).train()
    x = torch.randn(4, 3, 128, 128, device=dev)
    total = sum(p.numel() for p in net.parameters())
    for deferred in (False, True):
        net.zero_grad(set_to_none=True)
        ops.set_grad_chunks(4)
        ops.set_async_wgrad(deferred)
        try:
            ys = net(x)
            sum(y.square().mean() for y in ys).backward()
            n = int(ops.grad_chunk_count(net.grad_tag))
            assert n == 4
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                pieces = [ops.grad_chunk_wait(net.grad_tag, k) for k in range(n)]
            assert int(ops.grad_chunk_count(net.grad_tag)) == 0            # entry spent
            ops.wgrad_join()
        finally:
            ops.set_async_wgrad(False)
            ops.set_grad_chunks(0)
        torch.cuda.synchronize()
        assert sum(p.numel() for p in pieces) == total
        base = pieces[-1].data_ptr()
        ends = [p.data_ptr() + 4 * p.numel() for p in pieces]
        assert ends[0] == base + 4 * total
        for k in range(1, n):
            assert ends[k] == pieces[k - 1].data_ptr()                    # contiguous, descending
        pb = net.last_program
        off = base
        for p in pb.params:                                               # program order = buffer order
            assert p.grad.data_ptr() == off, 'gradient is not a view of the flat buffer'
            off += 4 * p.numel()
        flat = torch.cat([p.reshape(-1) for p in reversed(pieces)])
        want = torch.cat([p.grad.reshape(-1) for p in pb.params])
        assert torch.equal(flat, want) and bool(torch.isfinite(flat).all()) and float(flat.abs().max()) > 0
