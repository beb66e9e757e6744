"""Exactness of the training step.  With ``set_deterministic(True)`` every convolution weight gradient
runs on the fixed-order kernels of csrc/wgrad.hip (MIOpen's igemm_wrw...gkgs kernels accumulate with float atomics), and
at the BENCH shape (256 x 256, batch 32: MIOpen's forward / data-gradient choices are deterministic there; at 128 x 128 /
batch 8 its forward is not, tools/probes/determinism_forward.py) a whole stage-2 step -- two HRNets as encoder programs
on three streams, helper-thread backward, fused loss section, SGD -- is a pure function of its inputs:

  * two runs of the default runtime from the same seed: every parameter, every bank row, every loss bit-identical
    after three steps;
  * the same under a 1-rank nccl group with every collective of the N > 1 path forced on (RCCL all-gather, 9 in-place
    all-reduces per step in 5 launches, broadcasts): bit-identical to the run without a process group -- the comparison r02 could
    only make statistically (cosine >= 0.9);
  * the default runtime against plain autograd (module-by-module, one stream): both are reproducible but sum in
    different orders (batch-norm statistics from the convolution epilogue, gradient pairs added inside the
    normalisation backward), and ~150 stacked batch norms at the reference's std = 0.001 initialisation amplify that
    rounding; the FIRST update is compared per parameter (relative L2), not by a cosine over all of them."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(mode, steps=3):
    """mode: 'default' | 'plain' | 'rccl1' (1-rank nccl group, collectives forced)."""
    import bench
    from conftest import free_port
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.networks import hrnet
    dev = torch.device('cuda:0')
    glue = _lib.torch_glue()
    glue.set_deterministic(True)
    args = bench.make_args(32, 1024, 4096, 256, 'coco17', 'nccl', tempfile.mkdtemp(), steps + 1)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    plain = mode == 'plain'
    args.async_wgrad = not plain          # plain: no deferred weight gradients, no encoder programs (plain autograd)
    hrnet.ENCODER_PROGRAM = not plain
    if mode == 'rccl1':
        args.grad_sync = 'overlap'
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1, device_id=dev)
    try:
        tr = ContrastTrainer(args, force_collectives=mode == 'rccl1')
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        net = tr.unwrap(model)
        if plain:
            net.two_streams = 0
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        it = iter(data)
        losses, first = [], None
        for t in range(steps):
            losses.append(float(tr.train_step(next(it), model, contrast, opt, True)['loss']))
            if t == 0:
                torch.cuda.synchronize()
                first = {n: p.detach() - before[n] for n, p in net.named_parameters()}
        torch.cuda.synchronize()
        params = {n: p.detach().clone() for n, p in net.named_parameters()}
        banks = [b.clone() for b in contrast.banks()]
        launched = tr.grad_sync.launched if tr.grad_sync is not None else 0
    finally:
        hrnet.ENCODER_PROGRAM = True
        glue.set_async_wgrad(False)
        glue.set_grad_chunks(0)
        glue.set_deterministic(False)
        if dist.is_initialized():
            dist.destroy_process_group()
    return losses, params, banks, first, launched


@pytest.fixture(scope='module')
def default_run():
    return _run('default')


def _assert_identical(a, b):
    assert a[0] == b[0], (a[0], b[0])
    bad = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
    assert not bad, (len(bad), bad[:5])
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)


def test_default_runtime_is_bit_reproducible(default_run):
    _assert_identical(default_run, _run('default'))


def test_one_rank_rccl_group_is_bit_identical_to_no_group(default_run):
    """Every collective of the N > 1 path on RCCL (packed all-gather written by the heads kernel, 4 + 4 + 1 in-place
    all-reduces (as 4 coalesced pairs + 1) launched while the reverse loops still run, the broadcasts) is the identity with one rank: the run
    must reproduce the run without a process group BIT FOR BIT."""
    got = _run('rccl1')
    assert got[4] == 5, got[4]          # 4 coalesced chunk pairs + the rest bucket
    _assert_identical(default_run, got)


def test_plain_autograd_is_reproducible_and_its_first_update_matches_per_parameter(default_run):
    a = _run('plain')
    _assert_identical(a, _run('plain'))
    assert abs(a[0][0] - default_run[0][0]) <= 1e-5 * abs(a[0][0])              # same forward
    # first SGD update (lr * gradient): per parameter, relative L2
    worst, worst_name, big, num, den2, big_mass = 0.0, None, 0, 0.0, 0.0, 0.0
    for n, ua in a[3].items():
        ub = default_run[3][n]
        den = float(ua.norm())
        if den == 0:
            assert float(ub.norm()) == 0, n
            continue
        d2 = float((ua - ub).norm()) ** 2
        e = d2 ** 0.5 / den
        num, den2 = num + d2, den2 + den * den
        big += e > 1e-2
        big_mass += den * den if e > 1e-2 else 0.0
        if e > worst:
            worst, worst_name = e, n
    whole = (num / den2) ** 0.5
    print('worst per-parameter relative L2 of the first update: %.3e (%s); %d of %d above 1e-2 (%.2e of the update\'s squared '
          'norm); whole update, relative L2: %.3e' % (worst, worst_name, big, len(a[3]), big_mass / den2, whole))
    # Every tensor within 5e-2.  How many sit above 1e-2 depends on the BOX (which solvers MIOpen's Find picks for the encoders'
    # convolutions): 12-20, 28 and -- r06, the round's last full-suite run -- 103 of 1888 were seen with the same library, and an
    # A/B of two libraries on one box gave the same 28 tensors to the last digit; they are small bias vectors in front of batch
    # norms (true gradient zero: what is compared is amplified rounding).  So the count is bounded loosely (10 % of the tensors) and
    # what is bounded tightly is where the update's norm is: the tensors above 1e-2 carry < 1e-4 of its squared norm (measured: 7e-10).
    assert worst < 5e-2 and big <= len(a[3]) // 10 and big_mass / den2 < 1e-4, (worst, worst_name, big, big_mass / den2, whole)
