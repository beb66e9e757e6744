"""BASELINE config 5 as BASELINE.json words it ("bf16 mixed precision second-stage"): ``--encoder_dtype bf16`` (the
reference's ``--amp`` flag selects it too; its own hook is apex fp16, learning/contrast_trainer.py:65-72,
options/train_options.py:16-19) runs the two HRNets under bf16 autocast -- bf16 convolutions, batch-norm statistics,
master weights, SGD and the whole loss section in fp32.  One step at B = 8, 128 x 128: the loss section is still exact
against the oracle on the maps it was given (whole-step checker), the step's loss agrees with the fp32 run's to the
bf16 tolerance (3e-2 relative: 8 mantissa bits through ~150 layers), every parameter moves and stays finite."""
import os
import sys
import tempfile

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _one_step(encoder_dtype):
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.learning.engine import RecordingEngine
    dev = torch.device('cuda:0')
    args = bench.make_args(8, 1024, 4096, 128, 'coco17', 'nccl', tempfile.mkdtemp(), 3, encoder_dtype=encoder_dtype)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    eng = RecordingEngine()
    tr = ContrastTrainer(args, engine=eng)
    tr.device = dev
    try:
        model, contrast, opt, data = bench.build(args, tr, dev)
        net = tr.unwrap(model)
        assert net.encoder_dtype == (torch.bfloat16 if encoder_dtype == 'bf16' else torch.float32)
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        it = iter(data)
        eng.armed = False
        tr.train_step(next(it), model, contrast, opt, True)
        eng.armed = True
        out = tr.train_step(next(it), model, contrast, opt, True)
        torch.cuda.synchronize()
        moved = sum(1 for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n]))
        finite = all(bool(torch.isfinite(p).all()) for p in net.parameters())
        dtypes = {p.dtype for p in net.parameters()}
    finally:
        _lib.torch_glue().set_async_wgrad(False)
    return float(out['loss']), eng.records, moved, len(before), finite, dtypes


def test_bf16_encoders_train_and_the_loss_section_stays_exact():
    from oracle.check_step import check_records
    loss16, recs, moved, total, finite, dtypes = _one_step('bf16')
    assert dtypes == {torch.float32}                         # master weights stay fp32
    assert finite and moved == total, (moved, total)
    assert [r['kind'] for r in recs] == ['section']
    rep = check_records(recs)                                # fp32 tolerances: the section itself is not mixed precision
    print(rep)
    loss32, *_ = _one_step('fp32')
    assert abs(loss16 - loss32) <= 3e-2 * abs(loss32), (loss16, loss32)


def test_amp_flag_selects_the_bf16_encoders():
    import bench
    from hcmoco_amd.pycontrast.options.train_options import TrainOptions
    import contextlib
    import io
    tmp = tempfile.mkdtemp()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = TrainOptions().parse(['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--amp',
                                    '--model_path', tmp, '--tb_path', tmp, '--synthetic'])
    assert opt.amp and opt.encoder_dtype == 'bf16' and '_amp_O2' in opt.model_name
