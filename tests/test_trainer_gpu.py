"""ContrastTrainer on the GPU: the default runtime (encoder programs, three streams, deferred backward on
helper threads, quiet first step) against the plain one (module-by-module, one stream, everything issued
inline by autograd) from the same seed: same losses, same first parameter update (by direction), same banks after two SGD steps.
These are the DEFAULT-mode comparisons (MIOpen's atomic weight-gradient kernels make a whole step non-reproducible there,
so they are statistical); their exact counterparts -- bit-identical runs in deterministic mode at the bench shape -- are
tests/test_exact_gpu.py."""
import os
import sys
import tempfile

import pytest
from conftest import free_port
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(plain, steps=2):
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.networks import hrnet
    dev = torch.device('cuda:0')
    args = bench.make_args(8, 1024, 4096, 128, 'coco17', 'nccl', tempfile.mkdtemp(), steps + 1)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    args.async_wgrad = not plain          # plain: no deferred weight gradients, no encoder programs (plain autograd)
    hrnet.ENCODER_PROGRAM = not plain
    try:
        tr = ContrastTrainer(args)
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        if plain:
            tr.unwrap(model).two_streams = 0
        it = iter(data)
        before = {n: p.detach().clone() for n, p in tr.unwrap(model).named_parameters()}
        losses = [float(tr.train_step(next(it), model, contrast, opt, True)['loss'])]
        torch.cuda.synchronize()
        params = {n: p.detach() - before[n] for n, p in tr.unwrap(model).named_parameters()}    # first update
        banks = [[b.clone() for b in contrast.banks()]]
        losses += [float(tr.train_step(next(it), model, contrast, opt, True)['loss']) for _ in range(steps - 1)]
        torch.cuda.synchronize()
        banks.append([b.clone() for b in contrast.banks()])
    finally:
        hrnet.ENCODER_PROGRAM = True
        _lib.torch_glue().set_async_wgrad(False)
    return losses, params, banks


def test_default_runtime_matches_plain_autograd():
    l_fast, p_fast, b_fast = _run(plain=False)
    l_ref, p_ref, b_ref = _run(plain=True)
    assert abs(l_fast[0] - l_ref[0]) <= 1e-4 * abs(l_ref[0]), (l_fast, l_ref)        # same forward
    # the second loss sits behind one SGD step at random-init scale (weights ~1e-3, huge gradients)
    assert abs(l_fast[1] - l_ref[1]) <= 2e-2 * abs(l_ref[1]), (l_fast, l_ref)
    # the parameter updates (lr * momentum-filtered gradients) point the same way; element-wise equality
    # is not available through ~150 batch-norm layers (see tests/test_glue_gpu.py::_grads_agree)
    from test_glue_gpu import _grads_agree
    _grads_agree(p_fast, p_ref, min_cos=0.99)
    # banks after step 1: written from features of IDENTICAL weights -> tight; after step 2 the features sit
    # behind one SGD step at random-init scale (chaotic, like the second loss): rows must still point the same way
    for a, b in zip(b_fast[0], b_ref[0]):
        assert (a.float() - b.float()).abs().max().item() <= 1e-3
    for a, b, b1 in zip(b_fast[1], b_ref[1], b_ref[0]):
        rows = (b != b1).any(1)
        assert bool(rows.any())
        assert (a[~rows].float() - b[~rows].float()).abs().max().item() <= 1e-3
        cos = torch.nn.functional.cosine_similarity(a[rows].float(), b[rows].float(), dim=1)
        assert float(cos.min()) >= 0.9, float(cos.min())


class CheckingEngine(object):
    """HipLossEngine that re-evaluates every call with the oracle on CPU copies of the SAME tensors (same
    negative indices, same sampled pixels): the training loop under test is the product's, the numbers it
    produces are checked where they are produced, so encoder noise cannot blur the comparison."""

    def __init__(self):
        from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
        from oracle.oracle_engine import OracleLossEngine
        self.hip, self.oracle = HipLossEngine(), OracleLossEngine()
        self.calls = {'bank': 0, 'fmap': 0, 'grad': 0, 'regimes': set()}

    @staticmethod
    def _cpu(t):
        return None if t is None else t.detach().cpu()

    def bank(self, contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index, use_depth=None, use_rgb=None,
             idx=None):
        from oracle import hcmoco_oracle as O
        c = self._cpu
        idx = contrast.multinomial.draw_with_positive(index, contrast.K + 1)
        banks0 = [c(b).float() for b in contrast.banks()]
        total, losses, accs = self.hip.bank(contrast, f1, f2, f3, index, all_f1, all_f2, all_f3, all_index,
                                            use_depth=use_depth, use_rgb=use_rgb, idx=idx)
        lo, ao, go, _ = O.bank_nce(banks0, c(idx), [c(f1), c(f2), c(f3)], contrast.T, c(use_depth), c(use_rgb))
        assert torch.allclose(c(losses), lo, rtol=1e-5, atol=1e-6), (losses, lo)
        assert torch.allclose(c(accs), ao, atol=1e-3), (accs, ao)
        for i, (bank, ax) in enumerate(zip(contrast.banks(), (all_f1, all_f2, all_f3))):
            ref = O.bank_update(banks0[i], c(ax), c(all_index), contrast.m)
            assert (c(bank).float() - ref).abs().max().item() <= 1e-6
        for i, f in enumerate((f1, f2, f3)):
            def hook(g, i=i):
                ref = go[i]
                if float(ref.norm()) == 0:
                    assert float(g.abs().max()) == 0
                else:
                    assert float((c(g) - ref).norm() / ref.norm()) < 1e-4
                self.calls['grad'] += 1
            f.register_hook(hook)
        self.calls['bank'] += 1
        self.calls['regimes'].add((use_depth is not None, use_rgb is not None))
        return total, losses, accs

    def fmap_sampled(self, branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                     use_depth, use_rgb, num_samples, temperature, sample_ind=None, keep=None):
        c = self._cpu
        h, w = branches1[0].shape[-2:]
        sample_ind, keep = self.hip.dense_samples(depth_mask, h, w, num_samples, use_depth)
        total, meters = self.hip.fmap_sampled(branches1, branches2, proj1, proj2, feat3, depth_mask, joints2d, joints_vis,
                                              use_depth, use_rgb, num_samples, temperature, sample_ind=sample_ind, keep=keep)
        import copy
        p1, p2 = copy.deepcopy(proj1).cpu(), copy.deepcopy(proj2).cpu()
        with torch.no_grad():       # reference data flow: merge_all_res + full 1x1 projection + oracle losses
            _, want = self.oracle.fmap_sampled([c(m) for m in branches1], [c(m) for m in branches2], p1, p2, c(feat3),
                                               c(depth_mask), c(joints2d), c(joints_vis), c(use_depth), c(use_rgb),
                                               num_samples, temperature, sample_ind=c(sample_ind), keep=c(keep))
        assert torch.allclose(c(meters), want, rtol=1e-5, atol=1e-6), (meters, want)
        self.calls['fmap'] += 1
        return total, meters


def _main_args(method, extra):
    tmp = tempfile.mkdtemp()
    return ['--method', method, '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18', '--in_channel_list', '3,3',
            '--batch_size', '8', '--nce_k', '1024', '--world-size', '1', '--dist-backend', 'nccl', '--synthetic',
            '--synthetic_n_data', '4096', '--synthetic_size', '128', '--synthetic_steps', '2', '--epochs', '1',
            '--print_freq', '1', '--save_freq', '1', '--model_path', tmp, '--tb_path', tmp, '--seed', '3',
            '--learning_rate', '0.01', '--modality_missing', '1', '--synthetic_ntu', '1',
            '--synthetic_p_rgb', '0.7'] + list(extra)


@pytest.fixture
def _pg_env(monkeypatch):
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(free_port()))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SLURM_PROCID'):
        monkeypatch.delenv(k, raising=False)
    yield
    from hcmoco_amd import _lib
    _lib.torch_glue().set_async_wgrad(False)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def test_stage1_loop_on_the_hip_engine_through_main(_pg_env):
    """BASELINE config 1's method (CMCRGBD2S, learning/contrast_trainer.py:532-640) on the MI355X through the
    real entry point: two steps with use_depth AND use_rgb masks (the `both_*` row-selection regime), the fused
    bank kernel's losses / accuracies / d/dx / momentum update checked against the oracle at every step."""
    import warnings
    from hcmoco_amd.pycontrast import main_contrast
    eng = CheckingEngine()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        outs, trainer, model, contrast = main_contrast.main(_main_args('CMCRGBD2S', []), engine=eng)
        torch.cuda.synchronize()
    # r05's GPU run warned "The AccumulateGrad node's stream does not match ..." here: the six loss meters carried a grad_fn
    # and the running averages kept step 1's graph (made on one stream, the quiet first step) alive into step 2 (side streams)
    assert not [str(w.message)[:120] for w in caught if 'AccumulateGrad' in str(w.message)]
    assert trainer.device.type == 'cuda' and trainer.args.mem == 'bank'
    assert len(outs) == 6 and all(v == v for v in outs)
    assert eng.calls['bank'] == 2 and eng.calls['grad'] == 6 and eng.calls['regimes'] == {(True, True)}
    ck = torch.load(os.path.join(trainer.args.model_folder, 'current.pth'), map_location='cpu')
    assert ck['sampler']['offset'] == 2 and set(ck['contrast']) == {'memory_1', 'memory_2', 'memory_3'}


def test_stage2_loop_on_the_hip_engine_through_main(_pg_env):
    """Stage 2 (:894-1039) through the entry point with the default runtime (sampled projection, encoder
    programs, deferred backward): bank terms against the oracle, and the nine feature-map meters against the
    reference data flow (merge_all_res + full 1x1 projection + oracle losses) on the same branch maps."""
    from hcmoco_amd.pycontrast import main_contrast
    eng = CheckingEngine()
    argv = _main_args('CMCJointsPri3DRGBD2S', ['--linear_feat_map', '1', '--pri3d_num_samples_per_image', '64'])
    outs, trainer, model, contrast = main_contrast.main(argv, engine=eng)
    torch.cuda.synchronize()
    assert len(outs) == 4 and outs[0] == outs[0]
    assert eng.calls['bank'] == 2 and eng.calls['fmap'] == 2 and eng.calls['regimes'] == {(True, False)}


def test_flat_parameter_sgd_is_the_same_update_and_keeps_the_reference_checkpoint_layout():
    """learning/flat_sgd.py: after the first step every encoder parameter is a view of one flat tensor, the step it
    took is bit-identical to torch.optim.SGD on the separate tensors with the same gradients, and state_dict() is
    the per-parameter layout a plain SGD over model.parameters() loads."""
    import copy
    from hcmoco_amd.pycontrast.learning.flat_sgd import FlatParamSGD
    from hcmoco_amd.pycontrast.networks.hrnet import HighResolutionNet
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = HighResolutionNet(18).to(dev).train()
    head = torch.nn.Linear(18, 4).to(dev)
    model = torch.nn.ModuleDict({'enc': net, 'head': head})
    kw = dict(lr=0.03, momentum=0.9, weight_decay=1e-4)
    opt = FlatParamSGD(torch.optim.SGD(model.parameters(), fused=True, **kw), model)
    twin = copy.deepcopy(model)
    ref = torch.optim.SGD(twin.parameters(), fused=True, **kw)
    x = torch.randn(2, 3, 64, 64, device=dev)
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        maps = net(x)
        loss = sum(m.square().mean() for m in maps) + head(maps[0].mean((2, 3))).sum()
        loss.backward()
        from hcmoco_amd import _lib
        _lib.torch_glue().wgrad_join()
        for p, q in zip(model.parameters(), twin.parameters()):       # the twin takes the SAME gradients
            q.grad = None if p.grad is None else p.grad.detach().clone()
        opt.step()
        ref.step()
        assert len(opt._flat) == 1 and opt._flat[0][1].numel() == sum(p.numel() for p in net.last_program.params)
        for p, q in zip(model.parameters(), twin.parameters()):
            assert torch.equal(p, q)
        for p in net.last_program.params:
            assert p.untyped_storage().data_ptr() == opt._flat[0][1].untyped_storage().data_ptr()
    assert len(opt.param_groups[0]['params']) == 1 + sum(1 for p in model.parameters()
                                                          if all(p is not q for q in net.last_program.params))
    sd, rd = opt.state_dict(), ref.state_dict()
    assert sd['param_groups'][0]['params'] == rd['param_groups'][0]['params']
    assert sorted(sd['state']) == sorted(rd['state'])
    for k in rd['state']:
        assert torch.equal(sd['state'][k]['momentum_buffer'], rd['state'][k]['momentum_buffer'])
    # a plain optimizer loads it; the wrapper loads a plain optimizer's
    fresh = torch.optim.SGD(copy.deepcopy(twin).parameters(), fused=True, **kw)
    fresh.load_state_dict(sd)
    opt.load_state_dict(rd)
    sd2 = opt.state_dict()
    for k in rd['state']:
        assert torch.equal(sd2['state'][k]['momentum_buffer'], rd['state'][k]['momentum_buffer'])


def test_flat_parameter_sgd_resumes_from_a_reference_layout_checkpoint():
    """A per-parameter optimizer state loaded BEFORE the first step (resume_model's order) moves into the flat
    momentum buffer when the parameters are flattened: the next steps equal a plain SGD that simply continued."""
    import copy
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.flat_sgd import FlatParamSGD
    from hcmoco_amd.pycontrast.networks.hrnet import HighResolutionNet
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    net = HighResolutionNet(18).to(dev).train()
    kw = dict(lr=0.03, momentum=0.9, weight_decay=1e-4)
    plain = torch.optim.SGD(net.parameters(), fused=True, **kw)
    x = torch.randn(2, 3, 64, 64, device=dev)

    def backward(n):
        sum(m.square().mean() for m in n(x)).backward()
        _lib.torch_glue().wgrad_join()

    for _ in range(2):                                   # two plain steps: momentum buffers exist
        plain.zero_grad(set_to_none=True)
        backward(net)
        plain.step()
    twin = copy.deepcopy(net)
    cont = torch.optim.SGD(twin.parameters(), fused=True, **kw)
    cont.load_state_dict(copy.deepcopy(plain.state_dict()))
    flat = FlatParamSGD(torch.optim.SGD(net.parameters(), fused=True, **kw), net)
    flat.load_state_dict(copy.deepcopy(plain.state_dict()))            # before its first step
    for _ in range(2):
        flat.zero_grad(set_to_none=True)
        backward(net)
        for p, q in zip(net.parameters(), twin.parameters()):
            q.grad = p.grad.detach().clone()
        flat.step()
        cont.step()
        assert len(flat._flat) == 1
        for p, q in zip(net.parameters(), twin.parameters()):
            assert torch.equal(p, q)
    a, b = flat.state_dict(), cont.state_dict()
    for k in b['state']:
        assert torch.equal(a['state'][k]['momentum_buffer'], b['state'][k]['momentum_buffer'])
    # parameters re-allocated behind the optimizer's back (model.to / .float()): re-bound at the next step
    for p in net.parameters():
        p.data = p.data.clone()
    flat.zero_grad(set_to_none=True)
    backward(net)
    for p, q in zip(net.parameters(), twin.parameters()):
        q.grad = p.grad.detach().clone()
    flat.step()
    cont.step()
    for p, q in zip(net.parameters(), twin.parameters()):
        assert torch.equal(p, q)
    assert net.last_program.params[0].data_ptr() == flat._flat[0][1].data_ptr()
