"""The memory plugin (`CMCMem3`, `AliasMethod`, `CMCMoCo`, `build_mem`) on the GPU, driven the way
the reference trainer drives it (memory/mem_bank.py:172-205), against the golden vectors."""
import argparse

import pytest
import torch

from oracle import hcmoco_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def make_mem(g, d):
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    mem = CMCMem3(g['D'], g['n'], g['K'], g['T'], g['m']).to(d)
    mem.load_state_dict({'memory_%d' % i: g['bank0_%d' % i] for i in (1, 2, 3)})
    return mem


def test_forward_api_contract_vs_golden(golden):
    """Same signature / return tuple as the reference: six [B,K+1] logit tensors + labels, banks
    mutated in place after the reads, duplicates resolved last-wins."""
    g = golden('bank_nce')
    d = dev()
    mem = make_mem(g, d)
    xs = [g['x%d' % i].to(d).requires_grad_(True) for i in (1, 2, 3)]
    out = mem(xs[0], xs[1], xs[2], g['y'].to(d), g['all_x1'].to(d), g['all_x2'].to(d), g['all_x3'].to(d),
              g['all_y'].to(d), idx=g['idx'].to(d))
    assert len(out) == 7 and out[-1].dtype == torch.long and int(out[-1].abs().sum()) == 0
    for p in range(6):
        assert out[p].shape == (g['B'], g['K'] + 1)
        assert torch.allclose(out[p].cpu(), g['logits%d' % p], atol=1e-5)
    for i in (1, 2, 3):
        assert torch.allclose(getattr(mem, 'memory_%d' % i).cpu(), g['bank1_%d' % i], rtol=1e-6, atol=1e-7)
    # backward runs AFTER the in-place update and must still use the rows the logits were made from
    crit = torch.nn.CrossEntropyLoss()
    sum(crit(out[p], out[-1]) for p in range(6)).backward()
    for i in range(3):
        ref = g['none_gx%d' % (i + 1)]
        assert float((xs[i].grad.cpu() - ref).norm() / ref.norm()) < 1e-4


def test_forward_loss_draws_its_own_negatives_and_updates(golden):
    g = golden('bank_nce')
    d = dev()
    mem = make_mem(g, d)
    xs = [g['x%d' % i].to(d) for i in (1, 2, 3)]
    before = [b.clone() for b in mem.banks()]
    seed, off = mem.multinomial.seed, mem.multinomial.offset
    total, losses, accs = mem.forward_loss(xs[0], xs[1], xs[2], g['y'].to(d))
    # the negatives it drew are exactly the oracle's Philox stream for (seed, offset)
    idx = O.alias_draw_philox(mem.multinomial.prob.cpu(), mem.multinomial.alias.cpu(), g['B'] * (g['K'] + 1), seed, off)
    idx = idx.view(g['B'], g['K'] + 1).clone()
    idx[:, 0] = g['y']
    lo, ao, _, _ = O.bank_nce([b.cpu() for b in before], idx, [x.cpu() for x in xs], g['T'])
    assert torch.allclose(losses.cpu(), lo, rtol=1e-5, atol=1e-6) and torch.allclose(accs.cpu(), ao, atol=1e-3)
    assert mem.multinomial.offset == off + 1
    changed = (mem.memory_1 != before[0]).any(1).nonzero().flatten().cpu().tolist()
    assert sorted(changed) == sorted(set(g['y'].tolist()))       # without all_*: update from (x, y)


def test_alias_method_draw_surface():
    from hcmoco_amd.pycontrast.memory.alias_multinomial import AliasMethod
    am = AliasMethod(torch.ones(1000), seed=42)
    am.cuda()
    a = am.draw(5000)
    am2 = AliasMethod(torch.ones(1000), seed=42).to(dev())
    assert torch.equal(a, am2.draw(5000)) and a.shape == (5000,) and int(a.min()) >= 0 and int(a.max()) < 1000
    assert not torch.equal(a, am.draw(5000))                      # the offset advanced


def test_build_mem_and_moco_module(golden):
    from hcmoco_amd.pycontrast.memory.build_memory import build_mem
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    from hcmoco_amd.pycontrast.memory.mem_moco import CMCMoCo
    opt = argparse.Namespace(mem='bank+jointspri3d', feat_dim=128, nce_k=32, nce_t=0.07, nce_m=0.5, modal='RGBD2S')
    assert isinstance(build_mem(opt, 64), CMCMem3)
    opt.bank_dtype = 'bf16'
    assert build_mem(opt, 64).memory_1.dtype == torch.bfloat16
    opt.mem, opt.modal = 'moco', 'CMC'
    assert isinstance(build_mem(opt, 64), CMCMoCo)
    with pytest.raises(NotImplementedError):
        opt.mem = 'queue'
        build_mem(opt, 64)

    g = golden('moco_queue')
    d = dev()
    moco = CMCMoCo(g['D'], g['K'], g['T']).to(d)
    moco.load_state_dict({'memory_1': g['queue0_1'], 'memory_2': g['queue0_2']})
    for s in range(3):
        q1 = g['s%d_q1' % s].to(d).requires_grad_(True)
        l1, l2, lab = moco(q1, g['s%d_k1' % s].to(d), g['s%d_q2' % s].to(d), g['s%d_k2' % s].to(d),
                           all_k1=g['s%d_all_k1' % s].to(d), all_k2=g['s%d_all_k2' % s].to(d))
        assert torch.allclose(l1.cpu(), g['s%d_logits1' % s], atol=1e-5)
        assert torch.allclose(l2.cpu(), g['s%d_logits2' % s], atol=1e-5)
        assert moco.index == g['s%d_index' % s]
        assert torch.equal(moco.memory_1.cpu(), g['s%d_queue_1' % s])
        # d logits / d q through the library GEMM backward
        gl = torch.randn_like(l1)
        gq, = torch.autograd.grad(l1, q1, gl)
        ref = (gl[:, :1].cpu() * g['s%d_k2' % s] + gl[:, 1:].cpu() @ (g['queue0_2'] if s == 0 else g['s%d_queue_2' % (s - 1)])) / g['T']
        assert float((gq.cpu() - ref).norm() / ref.norm()) < 1e-5


def test_update_reads_chunk_views_of_one_feature_matrix(golden):
    """The trainer hands the module three column chunks of ONE [B, 384] matrix (torch.chunk views, stride
    384).  Each bank must be updated from ITS chunk: the contiguous copies made for the C ABI are distinct
    buffers that stay alive until the launch (a freed temporary is re-used by the next one otherwise, and all
    three banks would silently receive the last chunk)."""
    g = golden('bank_nce')
    d = dev()
    f = torch.cat([g['all_x1'], g['all_x2'], g['all_x3']], dim=1).to(d)
    y = g['all_y'].to(d)
    B = g['B']
    ref = make_mem(g, d)
    ref.forward_loss(g['all_x1'][:B].to(d), g['all_x2'][:B].to(d), g['all_x3'][:B].to(d), y[:B],
                     g['all_x1'].to(d), g['all_x2'].to(d), g['all_x3'].to(d), y, idx=g['idx'].to(d))
    mem = make_mem(g, d)
    c1, c2, c3 = torch.chunk(f, 3, dim=1)
    assert not c1.is_contiguous()
    mem.forward_loss(c1[:B], c2[:B], c3[:B], y[:B], c1, c2, c3, y, idx=g['idx'].to(d))
    torch.cuda.synchronize()
    for i in (1, 2, 3):
        assert torch.equal(getattr(mem, 'memory_%d' % i), getattr(ref, 'memory_%d' % i))
        assert torch.allclose(getattr(mem, 'memory_%d' % i).cpu(), g['bank1_%d' % i], rtol=1e-6, atol=1e-7)


def test_out_of_range_indices_are_contained_and_reported(golden):
    """A dataset index outside [0, n_data) (the reference would device-assert in index_select/index_copy_):
    nothing outside the banks is touched, the step still runs, and check_indices() raises."""
    g = golden('bank_nce')
    d = dev()
    mem = make_mem(g, d)
    B, n = g['B'], g['n']
    xs = [g['x%d' % i].to(d) for i in (1, 2, 3)]
    y = g['y'].to(d).clone()
    mem.forward_loss(xs[0], xs[1], xs[2], y)
    mem.check_indices()                                   # in range: silent
    guard = torch.full((1 << 20,), 7.0, device=d)         # memory next to whatever the allocator hands out
    y[1] = n + 12345
    y[2] = -3
    total, losses, accs = mem.forward_loss(xs[0], xs[1], xs[2], y)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(losses).all()) and bool((guard == 7.0).all())
    with pytest.raises(IndexError, match='outside'):
        mem.check_indices()
    mem.check_indices()                                   # flag is consumed
    bad = g['idx'].to(d).clone()
    bad[0, 5] = n
    mem(xs[0], xs[1], xs[2], g['y'].to(d), idx=bad)
    with pytest.raises(IndexError):
        mem.check_indices()
