"""The reference's MoCo loop, pinned (SURVEY 8a secondary row).  ``tests/golden/trace_moco.npz`` is a
4-step trace of the reference's OWN ``_train_moco`` / ``_shuffle_bn`` / ``momentum_update``
(learning/contrast_trainer.py:255-389, :167-210, :1041-1045) with the reference ``CMCMoCo`` (memory/mem_moco.py:91-142),
``torch.optim.SGD`` and the stand-in CMC encoder pair of tests/golden/standin.py (B = 6, K = 20: the ring pointer wraps).
This repo's ``ContrastTrainer._train_moco`` replays it with the recorded shuffle permutations injected:
  * CPU: queue = the oracle's functions (oracle/oracle_engine.py:OracleCMCMoCo);
  * GPU: queue = the product ``CMCMoCo`` on the HIP kernels hcm_moco_logits / hcm_moco_enqueue.
Checked per step: both logit sets (1e-5), losses / accuracies, the ring pointer and the queue contents (pointer
bit-exact, rows 1e-6 -- they are encoder outputs), query- and key-encoder weights after the step (2e-5 relative)."""
import argparse
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from conftest import load_golden  # noqa: E402


def _standin():
    spec = importlib.util.spec_from_file_location('standin', os.path.join(ROOT, 'tests', 'golden', 'standin.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _replay(device, make_mem, tol=1.0):
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    g = load_golden('trace_moco')
    standin = _standin()
    B, K, D, steps = int(g['B']), int(g['K']), int(g['D']), int(g['steps'])
    model, ema = standin.StandInMoCoEncoder(D=D), standin.StandInMoCoEncoder(D=D)
    as_t = lambda v: v if isinstance(v, torch.Tensor) else torch.tensor(v)      # 0-dim entries load as python numbers
    model.load_state_dict({k[3:]: as_t(v) for k, v in g.items() if k.startswith('w0_')})
    ema.load_state_dict({k[3:]: as_t(v) for k, v in g.items() if k.startswith('e0_')})
    model.to(device), ema.to(device)
    mem = make_mem(D, K, float(g['T']))
    with torch.no_grad():
        mem.memory_1.copy_(g['queue0_1'])
        mem.memory_2.copy_(g['queue0_2'])
    opt = torch.optim.SGD(model.parameters(), lr=float(g['lr']), momentum=float(g['momentum']),
                          weight_decay=float(g['weight_decay']))
    args = argparse.Namespace(jigsaw=False, modal='CMC', alpha=float(g['alpha']), local_rank=0, node_rank=0,
                              print_freq=10 ** 6, warm=False, rank=0)
    tr = ContrastTrainer(args, engine=object())
    tr.device = torch.device(device)
    tr.inject_shuffle_ids = [g['s%d_shuffle_ids' % t] for t in range(steps)]
    checked = {'n': 0}
    real_update = ContrastTrainer.momentum_update

    def after_step(m, e, a):                      # the reference records at the same point (:372)
        real_update(m, e, a)
        t = checked['n']
        last = tr.last_moco
        for i, key in enumerate(('logits1', 'logits2')):
            d = (last['logits'][i].cpu() - g['s%d_%s' % (t, key)]).abs().max()
            assert float(d) <= 1e-5 * tol + 1e-5 * tol * float(g['s%d_%s' % (t, key)].abs().max()), (t, key, float(d))
        got = torch.stack(last['losses']).cpu()
        assert torch.allclose(got, g['s%d_losses' % t], rtol=1e-5 * tol, atol=1e-6 * tol), (t, got, g['s%d_losses' % t])
        assert torch.allclose(torch.stack([a_.reshape(()) for a_ in last['accs']]).cpu(), g['s%d_accs' % t], atol=1e-3)
        assert mem.index == int(g['s%d_index' % t])                                   # ring pointer: bit-exact
        for q, key in ((mem.memory_1, 'queue_1'), (mem.memory_2, 'queue_2')):
            assert torch.allclose(q.cpu(), g['s%d_%s' % (t, key)], rtol=1e-5 * tol, atol=1e-6 * tol), (t, key)
        for net, pre in ((m, 'w'), (e, 'e')):
            for k, v in net.state_dict().items():
                want = torch.as_tensor(g['s%d_%s_%s' % (t, pre, k)]).float()
                err = float((v.cpu().float() - want).norm()) / max(float(want.norm()), 1e-30)
                assert err <= 2e-5 * tol, (t, pre, k, err)             # relative L2 per tensor
        checked['n'] += 1
    tr.momentum_update = after_step
    batches = [[g['s%d_data0' % t], g['s%d_data1' % t]] for t in range(steps)]
    outs = tr._train_moco(1, batches, model, ema, mem, None, opt)
    assert checked['n'] == steps
    assert np.allclose(np.array(outs[:2]), np.asarray(g['epoch_outs'])[:2], rtol=1e-5)
    # the pointer wrapped inside the trace, and untouched rows of the last step stayed where they were
    assert int(g['s%d_index' % (steps - 1)]) == (steps * B) % K and steps * B > K


def test_moco_trace_on_the_oracle_queue():
    from oracle.oracle_engine import OracleCMCMoCo
    _replay('cpu', OracleCMCMoCo)


@pytest.mark.gpu
def test_moco_trace_on_the_hip_queue():
    from hcmoco_amd.pycontrast.memory.mem_moco import CMCMoCo
    # the stand-in's convolutions / batch norm run on MIOpen here and on the CPU in the trace: four SGD steps apart the
    # two drift by a few 1e-5 relative; the ring pointer and WHICH rows are written stay exact
    _replay('cuda:0', lambda D, K, T: CMCMoCo(D, K, T).to('cuda:0'), tol=10.0)


WORKER = r'''
import argparse, importlib.util, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %r)
from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
from oracle.oracle_engine import OracleCMCMoCo
spec = importlib.util.spec_from_file_location('standin', os.path.join(%r, 'tests', 'golden', 'standin.py'))
standin = importlib.util.module_from_spec(spec); spec.loader.exec_module(standin)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.manual_seed(3)
model, ema = standin.StandInMoCoEncoder(D=32), standin.StandInMoCoEncoder(D=32)
mem = OracleCMCMoCo(32, 40, 0.2)
args = argparse.Namespace(jigsaw=False, modal='CMC', alpha=0.9, local_rank=rank, node_rank=0, print_freq=10 ** 6,
                          warm=False, rank=rank, mem='moco')
tr = ContrastTrainer(args, engine=object())
tr.local_group = None
ContrastTrainer.momentum_update(model, ema, 0)
x = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(100 + rank))
torch.manual_seed(50 + rank)                         # different host generators: rank 0's permutation must win
k, all_k = tr._shuffle_bn(x, ema)
# reference values: every rank's keys from the whole node batch, encoded in shuffled slices
torch.save({'k': k, 'all_k': all_k, 'x': x}, os.path.join(%r, 'rank%%d.pt' %% rank))
dist.destroy_process_group()
'''


def test_shuffle_bn_two_ranks_gloo(tmp_path):
    """contrast_trainer.py:167-210 with two ranks: the keys come back un-shuffled (row i of k belongs to row i of the
    local crops), all_k is rank-major, and the encoder saw MIXED batches (its batch statistics differ from encoding
    the local batch alone)."""
    import subprocess
    from conftest import free_port
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % (ROOT, ROOT, str(tmp_path)))
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(free_port()), str(script)],
                         capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS='2'), timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    r0, r1 = (torch.load(tmp_path / ('rank%d.pt' % r)) for r in (0, 1))
    assert torch.equal(r0['all_k'], r1['all_k']) and r0['all_k'].shape == (8, 64)
    # replay on one process: rank 0's permutation (seed 50), both ranks' crops, the same momentum encoder
    standin = _standin()
    torch.manual_seed(3)
    _, ema = standin.StandInMoCoEncoder(D=32), standin.StandInMoCoEncoder(D=32)
    torch.manual_seed(3)
    model = standin.StandInMoCoEncoder(D=32)
    ema.load_state_dict(model.state_dict())
    ema.train()
    torch.manual_seed(50)
    perm = torch.randperm(8)
    node_x = torch.cat([r0['x'], r1['x']])
    with torch.no_grad():
        enc = [ema(node_x[perm[r * 4:(r + 1) * 4]], mode=1) for r in (0, 1)]
    all_k = torch.cat(enc)
    assert torch.allclose(r0['all_k'], all_k, atol=1e-6)
    rev = torch.argsort(perm)
    assert torch.allclose(r0['k'], all_k[rev[0:4]], atol=1e-6) and torch.allclose(r1['k'], all_k[rev[4:8]], atol=1e-6)
    with torch.no_grad():
        alone = ema(r0['x'], mode=1)
    assert not torch.allclose(r0['k'], alone, atol=1e-4)          # shuffled batches => other batch statistics
