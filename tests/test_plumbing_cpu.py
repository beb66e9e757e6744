"""BASELINE config 1 ("plumbing, no GPU") and the host logic around the kernels, on CPU:
the real entry point / trainer / model / synthetic source / checkpointing run end to end with an
oracle-backed loss engine injected by the test; the product engine itself must fail loudly on CPU."""
import os
import subprocess
import sys
import tempfile

import pytest
from conftest import free_port
import torch

from hcmoco_amd.pycontrast import main_contrast
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
from hcmoco_amd.pycontrast.learning.engine import HipLossEngine
from hcmoco_amd.pycontrast.learning.util import AverageMeter
from oracle.oracle_engine import OracleLossEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def base_args(tmp, method, extra=()):
    return ['--method', method, '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18', '--in_channel_list', '3,3',
            '--batch_size', '4', '--nce_k', '64', '--world-size', '1', '--dist-backend', 'gloo', '--synthetic',
            '--synthetic_n_data', '256', '--synthetic_size', '64', '--synthetic_steps', '2', '--epochs', '1',
            '--print_freq', '1', '--save_freq', '1', '--model_path', tmp, '--tb_path', tmp, '--seed', '3',
            '--learning_rate', '0.01'] + list(extra)


@pytest.fixture(autouse=True)
def _fresh_pg(monkeypatch):
    monkeypatch.setenv('MASTER_PORT', str(free_port()))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SLURM_PROCID'):
        monkeypatch.delenv(k, raising=False)
    yield
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def test_config1_stage1_runs_on_cpu_with_oracle_engine():
    tmp = tempfile.mkdtemp()
    outs, trainer, model, contrast = main_contrast.main(base_args(tmp, 'CMCRGBD2S'), engine=OracleLossEngine())
    assert len(outs) == 6 and all(torch.isfinite(torch.tensor(outs)))
    ck = torch.load(os.path.join(trainer.args.model_folder, 'current.pth'), map_location='cpu')
    assert set(ck) >= {'model', 'contrast', 'optimizer', 'epoch'} and set(ck) <= {'model', 'contrast', 'optimizer', 'epoch', 'sampler'}
    assert all(k.startswith('module.') for k in ck['model'])                  # reference checkpoint layout
    assert set(ck['contrast']) == {'memory_1', 'memory_2', 'memory_3'}
    assert os.path.exists(os.path.join(trainer.args.model_folder, 'ckpt_epoch_1.pth'))


def test_stage2_accepts_method_name_runs_and_updates_only_indexed_rows():
    tmp = tempfile.mkdtemp()
    argv = base_args(tmp, 'CMCJointsPri3DRGBD2S', ['--linear_feat_map', '1', '--modality_missing', '1',
                                                    '--pri3d_num_samples_per_image', '16', '--synthetic_steps', '1'])
    torch.manual_seed(3)
    outs, trainer, model, contrast = main_contrast.main(argv, engine=OracleLossEngine())
    assert trainer.args.mem == 'bank+jointspri3d' and len(outs) == 4 and outs[0] == outs[0]
    # bank rows changed only at the batch's indices (stage-1 -> 2 hand-off relies on it)
    torch.manual_seed(3)                       # replay main(): seed -> build_model -> build_mem
    from hcmoco_amd.pycontrast.memory.build_memory import build_mem
    from hcmoco_amd.pycontrast.networks.build_backbone import build_model
    build_model(trainer.args)
    fresh = build_mem(trainer.args, 256)
    data = trainer_data_indices(trainer)
    changed = (fresh.memory_1 != contrast.memory_1).any(dim=1).nonzero().flatten().tolist()
    assert sorted(changed) == sorted(data)
    # stage-2 resume from the stage-1 style checkpoint
    argv2 = argv + ['--resume', os.path.join(trainer.args.model_folder, 'current.pth'), '--epochs', '2']
    torch.distributed.destroy_process_group()
    outs2, trainer2, *_ = main_contrast.main(argv2, engine=OracleLossEngine())
    assert outs2 is not None


def trainer_data_indices(trainer):
    from hcmoco_amd.pycontrast.datasets.synthetic import build_synthetic_contrast_loader
    ds, _, _ = build_synthetic_contrast_loader(trainer.args, 'cpu', 0, 1)
    return ds.pool[0][1].tolist()


def test_product_engine_fails_loudly_without_gpu():
    tmp = tempfile.mkdtemp()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        main_contrast.main(base_args(tmp, 'CMCRGBD2S'))          # default engine = HIP kernels


def test_pack_unpack_features_bit_exact_index():
    f = torch.randn(5, 384)
    index = torch.tensor([0, 1, 2 ** 31 + 5, 2 ** 40 + 123456789, 131071])
    packed = ContrastTrainer.pack_features(f, index)
    assert packed.shape == (5, 386)
    f2, i2 = ContrastTrainer.unpack_features(packed.clone())
    assert torch.equal(f2, f) and torch.equal(i2, index)


def test_average_meter_defers_sync_and_lr_schedule():
    m = AverageMeter()
    m.update(torch.tensor(2.0), 4)
    m.update(4.0, 4)
    assert m.avg == 3.0 and m.val == 4.0
    import argparse
    from hcmoco_amd.pycontrast.learning.base_trainer import BaseTrainer
    args = argparse.Namespace(learning_rate=0.1, cosine=False, lr_decay_epochs=[2, 4], lr_decay_rate=0.1, epochs=10)
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    tr = BaseTrainer(args)
    assert abs(tr.adjust_learning_rate(opt, 2) - 0.1) < 1e-12 and abs(tr.adjust_learning_rate(opt, 3) - 0.01) < 1e-12
    assert abs(tr.adjust_learning_rate(opt, 5) - 0.001) < 1e-12


def test_dense_samples_follow_the_mask():
    mask = torch.zeros(3, 32, 32)
    mask[0, 8:16, 8:16] = 1
    mask[2, :, :4] = 1
    ind, keep = HipLossEngine.dense_samples(mask, 8, 8, 50)
    assert keep.tolist() == [1, 0, 1]
    rows, cols = ind[0] // 8, ind[0] % 8
    assert bool(((rows >= 2) & (rows < 4) & (cols >= 2) & (cols < 4)).all())
    assert bool((ind[2] % 8 == 0).all())
    _, keep0 = HipLossEngine.dense_samples(mask, 8, 8, 50, use_depth=torch.zeros(3))
    assert keep0.tolist() == [0, 0, 0]


WORKER = r'''
import os, sys, tempfile, torch
sys.path.insert(0, %r)
from hcmoco_amd.pycontrast import main_contrast
from oracle.oracle_engine import OracleLossEngine
rank = int(os.environ['RANK'])
tmp = tempfile.mkdtemp()
argv = ['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18',
        '--in_channel_list', '3,3', '--batch_size', '4', '--nce_k', '32', '--dist-backend', 'gloo', '--synthetic',
        '--synthetic_n_data', '128', '--synthetic_size', '64', '--synthetic_steps', '2', '--epochs', '1',
        '--linear_feat_map', '1', '--modality_missing', '1', '--pri3d_num_samples_per_image', '8',
        '--model_path', tmp, '--tb_path', tmp, '--seed', '5', '--print_freq', '100', '--grad_sync', %r]
outs, trainer, model, contrast = main_contrast.main(argv, engine=OracleLossEngine())
# parameters only: BatchNorm running statistics are per-replica by design (local batches)
w = torch.cat([p.detach().flatten().double() for p in trainer.unwrap(model).parameters()])
torch.save({'bank': [b.clone() for b in contrast.banks()], 'wsum': w.sum(), 'wabs': w.abs().sum(), 'w': w.float(),
            'launched': None if trainer.grad_sync is None else trainer.grad_sync.launched},
           os.path.join(%r, 'rank%%d.pt' %% rank))
'''


def _run_two_ranks(grad_sync):
    out = tempfile.mkdtemp()
    script = os.path.join(out, 'worker.py')
    with open(script, 'w') as f:
        f.write(WORKER % (ROOT, grad_sync, out))
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS='2')
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', port, script],
                         capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0, r1 = (torch.load(os.path.join(out, 'rank%d.pt' % r)) for r in (0, 1))
    for b0, b1 in zip(r0['bank'], r1['bank']):
        assert torch.equal(b0, b1)                                     # replicated banks stay bit-identical
    assert float(r0['wsum']) == float(r1['wsum']) and float(r0['wabs']) == float(r1['wabs'])
    assert torch.equal(r0['w'], r1['w'])
    return r0


def test_world_size_2_gloo_replicas_stay_identical():
    """N>1 path on CPU: packed all-gather -> the SAME rank-major bank update on every replica,
    all three banks broadcast at start, averaged gradients -> identical weights; the bucketed
    (DistributedDataParallel) and the flat (one all-reduce after backward, the ROCm default) gradient
    averaging give the same model."""
    ddp = _run_two_ranks('ddp')
    flat = _run_two_ranks('flat')
    for b0, b1 in zip(ddp['bank'], flat['bank']):
        assert torch.allclose(b0, b1, rtol=1e-5, atol=1e-6)
    assert abs(float(ddp['wsum']) - float(flat['wsum'])) <= 1e-6 * float(ddp['wabs'])
    assert abs(float(ddp['wabs']) - float(flat['wabs'])) <= 1e-6 * float(ddp['wabs'])
    # the overlapped schedule (learning/grad_sync.py: one async all-reduce per bucket, launched in
    # completion order, re-bound .grad views) must give BIT-identical weights and banks to the flat one
    over = _run_two_ranks('overlap')
    assert flat['launched'] == 1 and over['launched'] > 1
    assert torch.equal(over['w'], flat['w'])
    for b0, b1 in zip(over['bank'], flat['bank']):
        assert torch.equal(b0, b1)


EIGHT_WORKER = r"""
import argparse, importlib.util, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
from oracle.oracle_engine import OracleLossEngine
spec = importlib.util.spec_from_file_location('standin', os.path.join(%(root)r, 'tests', 'golden', 'standin.py'))
standin = importlib.util.module_from_spec(spec); spec.loader.exec_module(standin)
dist.init_process_group('gloo')
rank, W = dist.get_rank(), dist.get_world_size()
B, n, K, H, J, S, steps = %(B)d, %(n)d, %(K)d, 16, 16, 6, 2
torch.manual_seed(100 + rank)              # replicas start DIFFERENT: wrap_up / broadcast_memory must make them equal
model = standin.StandInEncoder()
mem = CMCMem3(128, n, K, 0.07, 0.5, seed=1 + rank)
opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
args = argparse.Namespace(modality_missing=1, arch='HRNet', pri3d_num_samples_per_image=S, temperature=0.07, amp=False,
                          mem='bank+jointspri3d', rank=rank, local_rank=rank, world_size=W, grad_sync=%(mode)r,
                          linear_feat_map=0)
tr = ContrastTrainer(args, engine=OracleLossEngine())
tr.device = torch.device('cpu')
model, _, opt = tr.wrap_up(model, None, opt)
tr.broadcast_memory(mem)
w0 = {k: v.detach().clone() for k, v in tr.unwrap(model).state_dict().items()}
bank0 = [b.clone() for b in mem.banks()]
batches = standin.make_batches(steps, B * W, n, H, J, seed=77)      # the GLOBAL batch, identical on every rank
batches[0][1][2 * B + 1] = batches[0][1][5 * B + 2]                 # one bank row on ranks 2 AND 5: rank 5's sample wins
after = []
for t in range(steps):
    local = [item[rank * B:(rank + 1) * B] for item in batches[t]]
    out = tr.train_step(local, model, mem, opt, stage2=True)
    after.append([b.clone() for b in mem.banks()])
w = torch.cat([p.detach().flatten() for p in tr.unwrap(model).parameters()])
torch.save({'w0': w0, 'bank0': bank0, 'after': after, 'w': w, 'index0': batches[0][1], 'loss': float(out['loss']),
            'launched': None if tr.grad_sync is None else tr.grad_sync.launched},
           os.path.join(%(out)r, 'rank%%d.pt' %% rank))
"""


def _run_ranks(world, mode, B=3, n=96, K=12):
    out = tempfile.mkdtemp()
    script = os.path.join(out, 'worker.py')
    with open(script, 'w') as f:
        f.write(EIGHT_WORKER % dict(root=ROOT, mode=mode, out=out, B=B, n=n, K=K))
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                          '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script],
                         capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS='1'), timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return [torch.load(os.path.join(out, 'rank%d.pt' % r)) for r in range(world)]


def test_world_size_8_bank_update_is_the_oracle_update_of_the_rank_major_concatenation():
    """First-run readiness of the 8-GPU job, on CPU over gloo with the stand-in encoder and the
    oracle engine injected, in all three --grad_sync modes:
      * replicas that start from DIFFERENT weights and banks are equal after wrap_up / broadcast_memory and stay
        bit-identical (weights, all three banks) through two stage-2 steps;
      * the banks after step 1 equal ``oracle.bank_update`` of the RANK-MAJOR concatenation of all ranks' features and
        indices -- i.e. what ONE process computes on the global batch (no BatchNorm in the stand-in, so a sample's features
        do not depend on who else is in its batch) -- including a bank row that ranks 2 and 5 both write: the later
        rank's sample wins (memory/mem_bank.py:15-28 ``index_copy_`` on the gathered batch, learning/contrast_trainer.py:160-165
        rank order = concatenation order, :950-951);
      * flat, overlap and DistributedDataParallel agree to round-off (bit-identity of flat and overlap is a 2-rank property:
        a ring adds eight contributions in a buffer-position-dependent order)."""
    import importlib.util
    from oracle import hcmoco_oracle as O
    spec = importlib.util.spec_from_file_location('standin', os.path.join(ROOT, 'tests', 'golden', 'standin.py'))
    standin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(standin)
    W, B = 8, 3
    runs = {}
    for mode in ('flat', 'overlap', 'ddp'):
        ranks = _run_ranks(W, mode)
        r0 = ranks[0]
        for r in ranks[1:]:
            assert all(torch.equal(r['w0'][k], r0['w0'][k]) for k in r0['w0'])            # broadcast at start-up
            assert all(torch.equal(a, b) for a, b in zip(r['bank0'], r0['bank0']))        # all THREE banks
            assert torch.equal(r['w'], r0['w']), mode
            for t in range(2):
                assert all(torch.equal(a, b) for a, b in zip(r['after'][t], r0['after'][t])), (mode, t)
        # ---- one process on the global batch: features of the 24 samples from the broadcast weights, rank-major
        index = r0['index0']
        assert int(index[2 * B + 1]) == int(index[5 * B + 2])
        batches = standin.make_batches(2, B * W, 96, 16, 16, seed=77)
        ref = standin.StandInEncoder()
        ref.load_state_dict(r0['w0'])
        with torch.no_grad():
            f = ref(batches[0][0], batches[0][2])
        for i in range(3):
            want = O.bank_update(r0['bank0'][i], f[:, 128 * i:128 * (i + 1)], index, 0.5)
            got = r0['after'][0][i]
            touched = torch.zeros(96, dtype=torch.bool)
            touched[index] = True
            assert torch.equal(got[~touched], r0['bank0'][i][~touched])                   # untouched rows bit-identical
            assert torch.allclose(got, want, rtol=0, atol=1e-6), (mode, i, float((got - want).abs().max()))
            # the contested row holds rank 5's update, not rank 2's (they differ by far more than round-off)
            row = int(index[5 * B + 2])
            x5, x2 = f[5 * B + 2, 128 * i:128 * (i + 1)], f[2 * B + 1, 128 * i:128 * (i + 1)]
            upd = lambda x: torch.nn.functional.normalize(0.5 * r0['bank0'][i][row] + 0.5 * x, dim=0)
            assert float((got[row] - upd(x5)).abs().max()) < 1e-6 < 1e-3 < float((got[row] - upd(x2)).abs().max())
        runs[mode] = r0
    assert runs['flat']['launched'] == 1 and runs['overlap']['launched'] >= 1
    # flat vs overlap: bit-identical with TWO ranks (a + b is commutative; test_world_size_2_...).  With eight, a ring
    # all-reduce adds the eight contributions of an element in an order that depends on which eighth of the BUFFER the element
    # sits in, and the two modes cut the buffers differently: same sums, another association -- round-off, bounded here.
    d = (runs['flat']['w'] - runs['overlap']['w']).abs().max()
    print('8 ranks: max |flat - overlap| = %.3g, max |ddp - flat| = %.3g' % (
        float(d), float((runs['ddp']['w'] - runs['flat']['w']).abs().max())))
    assert torch.allclose(runs['flat']['w'], runs['overlap']['w'], rtol=1e-5, atol=1e-7)
    # banks: written in step 1 from features of identical weights (bit-identical), in step 2 from weights an ulp apart
    assert all(torch.equal(a, b) for a, b in zip(runs['flat']['after'][0], runs['overlap']['after'][0]))
    assert all(torch.allclose(a, b, rtol=0, atol=1e-6) for a, b in zip(runs['flat']['after'][1], runs['overlap']['after'][1]))
    assert torch.allclose(runs['ddp']['w'], runs['flat']['w'], rtol=1e-5, atol=1e-7)


UNUSED_WORKER = r'''
import os, sys, tempfile, torch
sys.path.insert(0, %r)
from hcmoco_amd.pycontrast import main_contrast
from oracle.oracle_engine import OracleLossEngine
rank = int(os.environ['RANK'])
tmp = tempfile.mkdtemp()
argv = ['--method', 'CMCRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18',
        '--in_channel_list', '3,3', '--batch_size', '4', '--nce_k', '32', '--dist-backend', 'gloo', '--synthetic',
        '--synthetic_n_data', '128', '--synthetic_size', '64', '--synthetic_steps', '2', '--epochs', '1',
        '--linear_feat_map', '1', '--model_path', tmp, '--tb_path', tmp, '--seed', '5', '--print_freq', '100',
        '--grad_sync', %r]
torch.manual_seed(5)
outs, trainer, model, contrast = main_contrast.main(argv, engine=OracleLossEngine())
net = trainer.unwrap(model)
names = [n for n, _ in net.named_parameters()]
if rank == 0:      # local rank 0 writes the checkpoint (contrast_trainer.py:117-140)
    ck = torch.load(os.path.join(trainer.args.model_folder, 'current.pth'), map_location='cpu')
    with_state = set(names[i] for i in ck['optimizer']['state'])
else:
    with_state = set(names[:101])
torch.save({'proj_grad_none': [p.grad is None for p in list(net.encoder1_linear.parameters()) + list(net.encoder2_linear.parameters())],
            'proj_has_momentum': [n for n in with_state if 'encoder1_linear' in n or 'encoder2_linear' in n],
            'n_state': len(with_state), 'proj': [p.detach().clone() for p in net.encoder1_linear.parameters()],
            'head': net.head1[0].weight.detach().clone()},
           os.path.join(%r, 'rank%%d.pt' %% rank))
'''


@pytest.mark.parametrize('mode', ['flat', 'overlap'])
def test_globally_unused_parameters_keep_no_gradient_with_two_ranks(mode):
    """Stage 1 with --linear_feat_map 1 never touches the two 1x1 projections.  With one GPU their ``.grad`` stays
    None and SGD skips them; averaging zeros into them across ranks would apply weight decay and momentum, i.e. the
    trained weights would depend on the GPU count (learning/grad_sync.py:_present).  Reference:
    DistributedDataParallel leaves globally unused parameters without a gradient (learning/contrast_trainer.py:74)."""
    out = tempfile.mkdtemp()
    script = os.path.join(out, 'worker.py')
    with open(script, 'w') as f:
        f.write(UNUSED_WORKER % (ROOT, mode, out))
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script],
                         capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS='2'), timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0, r1 = (torch.load(os.path.join(out, 'rank%d.pt' % r)) for r in (0, 1))
    for r in (r0, r1):
        assert r['proj_grad_none'] == [True] * 4 and r['proj_has_momentum'] == [] and r['n_state'] > 100
    assert torch.equal(r0['head'], r1['head'])
    # and they still hold their initial values (seed 5 -> build_model is the first consumer of the generator)
    import argparse
    from hcmoco_amd.pycontrast.networks.build_backbone import build_model
    torch.manual_seed(5)
    opt = argparse.Namespace(modal='RGBD2S', arch='HRNet', jigsaw=False, head='linear', feat_dim=128,
                             in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                             skeleton_meta_name='mpii', IN_Pretrain=None, depth_Pretrain=None, mem='bank')
    fresh, _ = build_model(opt)
    for a, b in zip(r0['proj'], fresh.encoder1_linear.parameters()):
        assert torch.equal(a, b.detach())


ASYM_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from hcmoco_amd.pycontrast.learning.grad_sync import GradSync
dist.init_process_group('gloo')
rank = dist.get_rank()
torch.manual_seed(0)
net = torch.nn.ModuleDict({'a': torch.nn.Linear(4, 3), 'b': torch.nn.Linear(4, 3), 'c': torch.nn.Linear(4, 3)})
params = list(net.parameters())
gs = GradSync(net, params, mode=%r)
x = torch.ones(2, 4) * (rank + 1)
# step: which branches each rank runs.  'c' is used by nobody at first, then by rank 1 ALONE (step 2), then by both;
# 'b' is dropped by rank 0 alone in step 1 (a shrinking local pattern: zeros, no collective)
plan = [({'a', 'b'}, {'a', 'b'}), ({'a'}, {'a', 'b'}), ({'a', 'b'}, {'a', 'b', 'c'}), ({'a', 'b', 'c'}, {'a', 'b', 'c'}),
        ({'a', 'b', 'c'}, {'a', 'b', 'c'})]
log = []
for step, use in enumerate(plan):
    for p in params:
        p.grad = None
    sum(net[k](x).sum() for k in sorted(use[rank])).backward()
    gs.reduce()
    log.append({'agreements': gs.agreements, 'launched': gs.launched,
                'grads': [None if p.grad is None else p.grad.clone() for p in params]})
torch.save(log, os.path.join(%r, 'rank%%d.pt' %% rank))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('mode', ['flat', 'overlap'])
def test_a_rank_whose_gradient_pattern_changes_alone_never_issues_a_collective_alone(mode):
    """ADVICE r03 (learning/grad_sync.py:_present): the presence all-reduce used to be gated on the LOCAL pattern, so
    a rank whose pattern changed alone entered a MAX all-reduce while its peer was in the bucket average (hang or
    silent corruption).  Now: the union is agreed at step 0 (every rank), a shrinking local pattern contributes zeros,
    a gradient outside the union is dropped for that step on the rank that has it, raises a flag that rides in the
    step's last bucket, and every rank re-agrees at the next step.  Two gloo ranks finish, count the same collectives,
    and hold identical gradients after every step.  Reference: DistributedDataParallel, contrast_trainer.py:74."""
    out = tempfile.mkdtemp()
    script = os.path.join(out, 'worker.py')
    with open(script, 'w') as f:
        f.write(ASYM_WORKER % (ROOT, mode, out))
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(free_port()), script],
                         capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS='2'), timeout=300)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0, r1 = (torch.load(os.path.join(out, 'rank%d.pt' % r)) for r in (0, 1))
    for a, b in zip(r0, r1):
        assert a['agreements'] == b['agreements'] and a['launched'] == b['launched']
        for ga, gb in zip(a['grads'], b['grads']):
            assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb))
    # params: a.w a.b b.w b.b c.w c.b ; x = 1 on rank 0, 2 on rank 1 -> d/dW = sum over the batch of x
    g = lambda step, i: r0[step]['grads'][i]
    assert g(0, 4) is None and torch.allclose(g(0, 0), torch.full((3, 4), 3.0))          # (2*1 + 2*2) / 2
    assert torch.allclose(g(1, 2), torch.full((3, 4), 2.0))                               # rank 0 sent zeros for b
    assert g(2, 4) is None                                  # rank 1's lone gradient for c: dropped, flag raised
    assert torch.allclose(g(3, 4), torch.full((3, 4), 3.0))  # union re-agreed by both ranks
    n = [r['agreements'] for r in r0]
    assert n[0] == n[1] == n[2] and n[3] > n[2] and n[4] == n[3]


def test_pretrain_handoff_stage1_to_stage2(capsys):
    """--pretrain strips the 7-char 'module.' prefix, loads matching keys, reports the rest and
    restores all three banks (main_contrast.py:52-67 of the reference)."""
    tmp = tempfile.mkdtemp()
    _, tr1, model1, contrast1 = main_contrast.main(base_args(tmp, 'CMCRGBD2S', ['--synthetic_steps', '1']),
                                                   engine=OracleLossEngine())
    ckpt = os.path.join(tr1.args.model_folder, 'current.pth')
    banks1 = [b.clone() for b in contrast1.banks()]
    w1 = model1.encoder1.conv1.weight.detach().clone()
    torch.distributed.destroy_process_group()
    argv = base_args(tmp, 'CMCJointsPri3DRGBD2S', ['--linear_feat_map', '1', '--modality_missing', '1',
                                                    '--pri3d_num_samples_per_image', '8', '--synthetic_steps', '1',
                                                    '--pretrain', ckpt, '--epochs', '0'])
    capsys.readouterr()
    _, tr2, model2, contrast2 = main_contrast.main(argv, engine=OracleLossEngine())     # epochs=0: load only
    out = capsys.readouterr().out
    assert 'Unmatched Keys: encoder1_linear.weight, encoder1_linear.bias, encoder2_linear.weight, encoder2_linear.bias' in out
    assert torch.equal(model2.encoder1.conv1.weight.detach(), w1)
    for a, b in zip(contrast2.banks(), banks1):
        assert torch.equal(a, b)


def test_transfer_ckpt_roundtrip_into_backbone():
    from hcmoco_amd.pycontrast import transfer_ckpt
    from hcmoco_amd.pycontrast.networks.hrnet import get_hrnet_w18_backbone
    tmp = tempfile.mkdtemp()
    _, tr, model, _ = main_contrast.main(base_args(tmp, 'CMCRGBD2S', ['--synthetic_steps', '1']),
                                         engine=OracleLossEngine())
    dst = os.path.join(tmp, 'rgb.pth')
    transfer_ckpt.main([os.path.join(tr.args.model_folder, 'current.pth'), dst, '--encoder', '2'])
    net = get_hrnet_w18_backbone()
    net.load_state_dict(torch.load(dst))                       # strict: every key present, none extra
    assert torch.equal(net.conv1.weight, model.encoder2.conv1.weight.detach().cpu())


def test_flat_param_sgd_without_encoder_programs_is_a_transparent_wrapper():
    """learning/flat_sgd.py on a model with no encoder program (CPU, or any module path): same parameters, same
    state_dict as the optimizer it wraps; lr edits through param_groups reach the inner optimizer."""
    import copy
    from hcmoco_amd.pycontrast.learning.flat_sgd import FlatParamSGD
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 2))
    t = copy.deepcopy(m)
    kw = dict(lr=0.1, momentum=0.9, weight_decay=1e-3)
    a = FlatParamSGD(torch.optim.SGD(m.parameters(), **kw), m)
    b = torch.optim.SGD(t.parameters(), **kw)
    x = torch.randn(7, 5)
    for i in range(3):
        for net, opt in ((m, a), (t, b)):
            opt.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            opt.step()
        a.param_groups[0]['lr'] = b.param_groups[0]['lr'] = 0.05
    for p, q in zip(m.parameters(), t.parameters()):
        assert torch.equal(p, q)
    sa, sb = a.state_dict(), b.state_dict()
    assert sa['param_groups'] == sb['param_groups']
    for k in sb['state']:
        assert torch.equal(sa['state'][k]['momentum_buffer'], sb['state'][k]['momentum_buffer'])
    a.load_state_dict(sb)
    assert a.inner.param_groups[0]['lr'] == 0.05


def test_cpu_affinity_helper_parses_sysfs_lists_and_is_a_no_op_without_a_gpu():
    """learning/affinity.py: the cpulist grammar of sysfs, and no change (None) where no GPU / sysfs entry exists."""
    import os
    from hcmoco_amd.pycontrast.learning import affinity
    assert affinity._parse_cpulist('64-127,192-255\n') == list(range(64, 128)) + list(range(192, 256))
    assert affinity._parse_cpulist('3') == [3] and affinity._parse_cpulist('') == []
    before = os.sched_getaffinity(0)
    assert affinity.pin_to_gpu_node(0) is None and affinity.pin_to_gpu_node(0, mode='0') is None
    assert os.sched_getaffinity(0) == before


def test_kept_pixels_are_the_ones_the_nearest_resize_reads():
    """CMC3HRNetSGCNPN2SingleHead.pts2depth_resized (networks/build_backbone.py:299-300 of the reference, fused) interpolates
    the depth feature map only at the pixels F.interpolate(size=..., mode='nearest') keeps; the index list must BE torch's
    rounding for every ratio, integer or not, and the side of the HRNet's first map must be what the stem produces."""
    import torch.nn.functional as F
    from hcmoco_amd.pycontrast.networks.build_backbone import CMC3HRNetSGCNPN2SingleHead as M
    for h, w, oh, ow in [(256, 256, 64, 64), (320, 320, 80, 80), (36, 52, 10, 13), (17, 9, 5, 3), (8, 8, 8, 8), (12, 20, 24, 40)]:
        x = torch.randn(2, 3, h, w)
        keep = M.kept_pixels(h, w, oh, ow, x.device)
        assert keep.shape == (oh * ow,) and int(keep.min()) >= 0 and int(keep.max()) < h * w
        assert torch.equal(x.reshape(2, 3, h * w).index_select(2, keep).reshape(2, 3, oh, ow), F.interpolate(x, size=(oh, ow)))
    stem = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, 2, 1, bias=False), torch.nn.Conv2d(4, 4, 3, 2, 1, bias=False))
    for n in (224, 256, 288, 320, 33, 7):
        assert stem(torch.zeros(1, 3, n, n + 2)).shape[-2:] == (M._stem_hw(n), M._stem_hw(n + 2))
