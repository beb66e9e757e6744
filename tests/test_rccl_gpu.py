"""The N>1 path executed on RCCL on the GPU box: a 1-rank ``nccl`` group on
cuda:0 with every collective forced on -- packed feature/index ``all_gather_into_tensor``, the three
bank ``broadcast``s, the model broadcast, and the chunked in-place ``all_reduce``s of the encoders'
flat gradient buffers that learning/grad_sync.py launches while the reverse loops are still being
issued.  With one rank every collective is the identity, so two training steps must reproduce the
run that has no process group at all.  (Reference: learning/contrast_trainer.py:74, 81-91, 160-165.)
Default mode: compared statistically where whole steps are involved; the bit-exact form of the same comparison
(deterministic mode, bench shape) is tests/test_exact_gpu.py.
"""
import os
import sys
import tempfile

import pytest
from conftest import free_port
import torch
import torch.distributed as dist

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNKS = 4


def _run(grad_sync, steps=2, stage2=True, check_identity=False):
    """grad_sync None: no process group (collectives skipped); 'overlap' / 'flat': 1-rank nccl group.
    check_identity: join the helper threads and snapshot every gradient BEFORE GradSync.reduce, compare
    bit for bit after it (one rank: every all-reduce must be the identity, whatever buffer it ran on)."""
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    dev = torch.device('cuda:0')
    args = bench.make_args(8, 1024, 4096, 128, 'coco17', 'nccl', tempfile.mkdtemp(), steps + 1)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    info = {}
    if grad_sync is not None:
        args.grad_sync = grad_sync
        os.environ['HCM_GRAD_CHUNKS'] = str(CHUNKS)
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(),
                                rank=0, world_size=1, device_id=dev)
    try:
        tr = ContrastTrainer(args, force_collectives=grad_sync is not None)
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        net = tr.unwrap(model)
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        if check_identity:
            sync = tr.grad_sync
            real = sync.reduce

            def checked(join=None):
                if join is not None:
                    join()
                torch.cuda.synchronize()
                snap = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
                n = real(None)
                torch.cuda.synchronize()
                bad = [k for k, p in net.named_parameters()
                       if (snap[k] is None and p.grad is not None)
                       or (snap[k] is not None and not torch.equal(p.grad, snap[k]))]
                info.setdefault('not_identity', []).extend(bad)
                info.setdefault('none_grads', []).append(sum(v is None for v in snap.values()))
                return n
            sync.reduce = checked
        it = iter(data)
        losses = []
        for _ in range(steps):
            out = tr.train_step(next(it), model, contrast, opt, stage2)
            losses.append(float(out['loss']))
            if tr.grad_sync is not None:
                info.setdefault('launched', []).append(tr.grad_sync.launched)
        torch.cuda.synchronize()
        if tr.grad_sync is not None and grad_sync == 'overlap':
            # the encoders' gradients were reduced in place: every .grad is a view of ONE buffer per encoder
            for enc in (net.encoder1, net.encoder2):
                ptrs = {p.grad.untyped_storage().data_ptr() for p in enc.parameters()}
                info.setdefault('storages', []).append(len(ptrs))
                info.setdefault('numel', []).append(sum(p.numel() for p in enc.parameters()))
                info.setdefault('storage_numel', []).append(
                    next(enc.parameters()).grad.untyped_storage().nbytes() // 4)
            info['groups'] = len(tr.grad_sync.groups)
        update = torch.cat([(p.detach() - before[n]).flatten() for n, p in net.named_parameters()]).double()
        banks = [b.clone() for b in contrast.banks()]
    finally:
        _lib.torch_glue().set_async_wgrad(False)
        _lib.torch_glue().set_grad_chunks(0)
        if dist.is_initialized():
            dist.destroy_process_group()
    return losses, update, banks, info


def _same(a, b):
    """Statistical agreement with the run that has no process group.  Individual parameters are NOT
    comparable between two runs of even the same configuration (MIOpen's kernels are not run-to-run
    deterministic and ~150 stacked batch-norm layers amplify it: batch-norm biases differ by tens of percent,
    the second loss by ~1e-3); a missing, doubled or mis-scaled reduction moves these numbers by far more."""
    la, ua, ba, _ = a
    lb, ub, bb, _ = b
    assert abs(la[0] - lb[0]) <= 1e-5 * abs(lb[0]), (la, lb)
    # (exactness lives in test_every_collective_is_the_identity... and test_two_ranks_on_one_gpu...; these bounds
    # only have to separate "same training run up to chaos" from "wrong / missing / doubled reduction")
    assert abs(la[1] - lb[1]) <= 2e-2 * abs(lb[1]), (la, lb)
    cos = float(torch.dot(ua, ub) / (ua.norm() * ub.norm()))
    ratio = float(ua.norm() / ub.norm())
    assert cos >= 0.9 and 0.8 <= ratio <= 1.25, (cos, ratio)
    # bank rows written in step 2 come from features behind one chaotic SGD step: direction only
    for x, y in zip(ba, bb):
        cos = torch.nn.functional.cosine_similarity(x.float(), y.float(), dim=1)
        assert float(cos.min()) >= 0.9, float(cos.min())
        assert float((cos < 0.9999).float().mean()) < 0.02       # and only the handful of rows of the last batch


@pytest.fixture(scope='module')
def baseline():
    return _run(None)


def test_every_collective_is_the_identity_in_a_one_rank_group():
    """Exactness: with the helper threads joined first, the gradients before and after GradSync.reduce are
    bit-identical -- chunk views alias the encoders' buffers, the rest buckets are copied in, reduced and
    re-bound without loss, parameters without a gradient on any rank keep ``.grad = None``."""
    got = _run('overlap', check_identity=True)
    info = got[3]
    assert info['not_identity'] == [], info['not_identity'][:10]
    assert info['storages'] == [1, 1] and info['storage_numel'] == info['numel'], info
    # 4 chunk PAIRS (the k-th chunks of the two HRNets go out as one RCCL launch) + the rest bucket
    assert info['launched'] == [CHUNKS + 1] * len(info['launched']), info


def test_overlapped_rccl_allreduce_one_rank_group(baseline):
    """The real schedule: all-reduces launched while the reverse loops are still being issued."""
    got = _run('overlap')
    info = got[3]
    # per step: CHUNKS in-place all-reduces per HRNet, the two encoders' k-th chunks coalesced into one launch, + one for
    # the remaining parameters; the first step (quiet Find, everything inline) takes the same route
    assert info['storages'] == [1, 1], info                    # one flat buffer per encoder ...
    assert info['storage_numel'] == info['numel'], info        # ... and it is dense (nothing but gradients)
    # 4 chunk PAIRS (the k-th chunks of the two HRNets go out as one RCCL launch) + the rest bucket
    assert info['launched'] == [CHUNKS + 1] * len(info['launched']), info
    _same(got, baseline)


def test_flat_rccl_allreduce_one_rank_group(baseline):
    got = _run('flat')
    assert got[3]['launched'] == [1, 1]
    _same(got, baseline)


TWO_RANK_WORKER = r'''
import os, sys, tempfile, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import bench
from hcmoco_amd import _lib
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)                       # both ranks on the box's one GPU
dev = torch.device('cuda:0')
dist.init_process_group('gloo', rank=rank, world_size=world)
os.environ['HCM_GRAD_CHUNKS'] = '4'
args = bench.make_args(8 * world, 1024, 4096, 128, 'coco17', 'gloo', tempfile.mkdtemp(), 4)
args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = rank, world, rank, 0, False
args.grad_sync = %r
tr = ContrastTrainer(args)
tr.device = dev
model, contrast, opt, data = bench.build(args, tr, dev)
torch.cuda.manual_seed(1234 + rank)
it = iter(data)
launched = []
for _ in range(3):
    out = tr.train_step(next(it), model, contrast, opt, True)
    launched.append(tr.grad_sync.launched)
torch.cuda.synchronize()
_lib.torch_glue().set_async_wgrad(False)
w = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
torch.save({'w': w, 'banks': [b.cpu() for b in contrast.banks()], 'launched': launched, 'loss': float(out['loss']),
            'index': data.pool[0][1].cpu()}, os.path.join(%r, 'rank%%d.pt' %% rank))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('mode', ['overlap', 'flat'])
def test_two_ranks_on_one_gpu_stay_bit_identical(mode, tmp_path):
    """Two replicas (different data, different negatives) sharing cuda:0, collectives over gloo on device
    tensors, three steps with the DEFAULT runtime: helper threads issue the reverse loops while GradSync
    launches the chunk all-reduces behind the chunk events.  Every rank must end with bit-identical parameters
    and banks: a chunk reduced before its last layer was written, a gradient re-bound to the wrong buffer or a
    bank update that is not rank-major would make the replicas drift.  (RCCL refuses two ranks on one device,
    so the multi-rank data path is exercised with gloo here and RCCL with one rank above.)"""
    import subprocess
    script = tmp_path / 'worker.py'
    script.write_text(TWO_RANK_WORKER % (ROOT, mode, str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS='4')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(free_port()), str(script)],
                         capture_output=True, text=True, env=env, timeout=800)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    r0, r1 = (torch.load(tmp_path / ('rank%d.pt' % r)) for r in (0, 1))
    assert torch.equal(r0['w'], r1['w'])
    for a, b in zip(r0['banks'], r1['banks']):
        assert torch.equal(a, b)
    assert not torch.equal(r0['index'], r1['index'])                       # the ranks did train on different samples
    assert r0['loss'] == r0['loss'] and r0['loss'] != r1['loss']
    if mode == 'overlap':
        assert r0['launched'] == [2 * 4 + 1] * 3, r0['launched']         # gloo: no coalescing, 4 chunks x 2 HRNets + the rest
    else:
        assert r0['launched'] == [1, 1, 1]


def test_hrnetpn_under_a_one_rank_rccl_group(tmp_path):
    """Config 4's model (HRNet + PointNet++ + SemGCN) through the N>1 control path on RCCL: one HRNet goes chunk by
    chunk, the PointNet++ shared MLPs (deferred weight gradients on their own stream) are joined before they
    travel in the rest bucket.  One rank => every collective is the identity => gradients unchanged by reduce()."""
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    dev = torch.device('cuda:0')
    args = bench.make_args(4, 1024, 4096, 64, 'coco17', 'nccl', str(tmp_path), 3, arch='HRNetPN', width=18)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    args.grad_sync = 'overlap'
    os.environ['HCM_GRAD_CHUNKS'] = str(CHUNKS)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1,
                            device_id=dev)
    try:
        tr = ContrastTrainer(args, force_collectives=True)
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        sync, bad, launched = tr.grad_sync, [], []
        real = sync.reduce

        def checked(join=None):
            if join is not None:
                join()
            torch.cuda.synchronize()
            snap = {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
            n = real(None)
            torch.cuda.synchronize()
            bad.extend(k for k, p in model.named_parameters()
                       if (snap[k] is None and p.grad is not None)
                       or (snap[k] is not None and not torch.equal(p.grad, snap[k])))
            launched.append(n)
            return n
        sync.reduce = checked
        it = iter(data)
        for _ in range(2):
            out = tr.train_step(next(it), model, contrast, opt, True)
        torch.cuda.synchronize()
        assert bad == [] and launched == [CHUNKS + 1, CHUNKS + 1], (bad[:5], launched)
        assert bool(torch.isfinite(out['loss']))
    finally:
        _lib.torch_glue().set_async_wgrad(False)
        _lib.torch_glue().set_grad_chunks(0)
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['exit', 'raise'])
def test_bench_line_survives_a_dead_rank(kind):
    """the N > 1 bench line cannot fail silently.  ``python bench.py --gpus 2 --backend gloo`` (two ranks on
    this box's one GPU), rank 1 dies in timed step 1 -- ``exit``: os._exit without a word (the launcher tears the job down,
    rank 0 hears SIGTERM while it sits in a collective); ``raise``: an exception (rank 1 leaves a note in the rendez-vous
    store).  Either way rank 0 prints ONE JSON line with ``error``, ``phase`` and the ``comm`` block, inside 150 s of the
    fault (the whole run, model build and MIOpen warm-up included, gets 400 s here)."""
    import json
    import subprocess
    import time
    env = dict(os.environ)
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '4',
                          '--warmup', '1', '--batch_per_gpu', '4', '--nce_k', '512', '--n_data', '2048', '--size', '64',
                          '--no_check', '--no_cpu_baseline', '--fault', '1:1:' + kind],
                         capture_output=True, text=True, env=env, timeout=400)
    took = time.time() - t0
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, (res.returncode, res.stdout[-1500:], res.stderr[-3000:])
    out = json.loads(lines[0])
    assert res.returncode != 0
    assert out['value'] is None and out['error'] and out['phase'] == 'timed steps', out
    assert out['n_gpus'] == 2 and out['comm']['world_size'] == 2 and out['comm']['ranks_seen'] == 2, out
    assert out['comm']['hsa_enable_ipc_mode_legacy']['value'] == '0'
    if kind == 'raise':
        assert 'injected fault' in out['error'], out['error']
    assert took < 400
