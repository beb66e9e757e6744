"""HRNetPN arch (BASELINE config 4): PointNet++ depth encoder driven by the HIP point ops.
CPU: state_dict parity with the reference, and the host-side PointNet++ modules run against the
oracle-backed native shim.  GPU: full HRNetPN stage-2 steps run (parity of the cloud encoder and of the whole
HRNetPN model with the reference: tests/test_pn_reference.py, tests/test_pn_reference_gpu.py)."""
import argparse

import numpy as np
import pytest
import torch

from hcmoco_amd.pycontrast.networks.build_backbone import build_model
from hcmoco_amd.pycontrast.networks.pointnet2 import pointnet2_utils
from hcmoco_amd.pycontrast.networks.pointnet2_msg import Pointnet2MSG
from oracle import pointnet2_shim


def pn_opt(skel='mpii'):
    return argparse.Namespace(modal='RGBD2S', arch='HRNetPN', jigsaw=False, head='linear', feat_dim=128,
                              in_channel_list=[3, 3], linear_feat_map=1, width=18, pool_method='mean',
                              skeleton_meta_name=skel, IN_Pretrain=None, depth_Pretrain=None, mem='bank')


def test_hrnetpn_state_dict_matches_reference():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'model_hrnetpn_w18_keys.npz'))
    model, _ = build_model(pn_opt())
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g['keys']]
    assert [str(list(v.shape)) for v in sd.values()] == [str(s) for s in g['shapes']]
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])


def cloud(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(B, N, 3, generator=g) * torch.tensor([1.0, 2.0, 0.6]) - torch.tensor([0.5, 1.0, 0.3])
    src = torch.randint(0, N, (B, N // 3), generator=g)          # duplicates, like sampling with replacement
    for b in range(B):
        pts[b, :N // 3] = pts[b, src[b]]
    return pts


def run_msg(device, native, pts, weights=None):
    old = pointnet2_utils.pointnet2
    pointnet2_utils.pointnet2 = native
    try:
        torch.manual_seed(0)
        net = Pointnet2MSG(input_channels=0)
        if weights is not None:
            net.load_state_dict(weights)
        net.to(device).train()
        x = pts.to(device).requires_grad_(False)
        out = net(x)
        loss = (out * torch.linspace(-1, 1, out.numel(), device=device).view_as(out)).sum()
        loss.backward()
        grads = {k: p.grad.detach().cpu() for k, p in net.named_parameters()}
        return out.detach().cpu(), grads, {k: v.detach().cpu() for k, v in net.state_dict().items()}
    finally:
        pointnet2_utils.pointnet2 = old


def test_pointnet2_msg_runs_on_cpu_with_oracle_shim():
    out, grads, _ = run_msg('cpu', pointnet2_shim, cloud(1, 4096, 1))
    assert out.shape == (1, 128, 4096) and torch.isfinite(out).all()
    assert all(torch.isfinite(g).all() for g in grads.values())


# (r06) HIP-vs-own-modules at 2e-3 lived here; superseded by tests/test_pn_reference_gpu.py: the default GPU runtime against
# fixtures recorded from the REFERENCE's modules, at the 1e-4 / 1e-5-of-max gate.


@pytest.mark.gpu
def test_hrnetpn_stage2_step_runs_on_gpu():
    from hcmoco_amd.pycontrast.datasets.synthetic import SyntheticContrastData
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.memory.mem_bank import CMCMem3
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    opt = pn_opt()
    model, _ = build_model(opt)
    model.to(dev).train()
    args = argparse.Namespace(arch='HRNetPN', modality_missing=1, pri3d_num_samples_per_image=32, temperature=0.07,
                              rank=0, local_rank=0, world_size=1)
    trainer = ContrastTrainer(args)
    trainer.device = dev
    data = SyntheticContrastData(512, 2, size=64, joints=16, steps=1, device=dev, pool=1, ntu=True)
    mem = CMCMem3(128, 512, 128, 0.07, 0.5).to(dev)
    sgd = torch.optim.SGD(model.parameters(), lr=0.01)
    before = mem.memory_2.clone()
    out = trainer.train_step(data.pool[0], model, mem, sgd, stage2=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out['loss']) and torch.isfinite(out['fmap']).all()
    changed = (mem.memory_2 != before).any(1).nonzero().flatten().tolist()
    assert sorted(changed) == sorted(data.pool[0][1].tolist())
    assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in model.encoder2.parameters())


@pytest.mark.gpu
def test_hrnetpn_w32_stage2_steps_at_256(tmp_path):
    """BASELINE config 4's model at its own width and resolution (HRNet-w32 RGB encoder + PointNet++ MSG on
    4096-point clouds + SemGCN, 256x256, K=16384; batch 8 to keep the test short): two stage-2 steps through
    bench.build / ContrastTrainer with the default runtime.  Losses finite and moving, banks change only at
    the batch's rows, every encoder receives gradients, the w32 HRNet ran as an encoder program."""
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    dev = torch.device('cuda:0')
    args = bench.make_args(8, 16384, 131072, 256, 'coco17', 'nccl', str(tmp_path), 3, arch='HRNetPN', width=32)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    try:
        tr = ContrastTrainer(args)
        tr.device = dev
        model, contrast, opt, data = bench.build(args, tr, dev)
        assert model.encoder1.width == 32 and sum(p.numel() for p in model.encoder1.parameters()) > 29e6
        before = [b.clone() for b in contrast.banks()]
        it = iter(data)
        rows, losses = [], []
        for _ in range(2):
            batch = next(it)
            rows += batch[1].tolist()
            out = tr.train_step(batch, model, contrast, opt, True)
            losses.append(float(out['loss']))
            assert bool(torch.isfinite(out['fmap']).all()) and bool(torch.isfinite(out['bank_losses']).all())
        torch.cuda.synchronize()
        contrast.check_indices()
        assert all(v == v for v in losses) and losses[0] != losses[1]
        assert model.encoder1.last_program is not None
        for bank, b0 in zip(contrast.banks(), before):
            changed = (bank != b0).any(1).nonzero().flatten().tolist()
            assert sorted(changed) == sorted(set(rows))
        for enc in (model.encoder1, model.encoder2, model.encoder3, model.encoder1_linear, model.encoder2_linear):
            assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in enc.parameters())
    finally:
        _lib.torch_glue().set_async_wgrad(False)


@pytest.mark.gpu
@pytest.mark.parametrize('width,size,B', [(18, 64, 4), (18, 96, 3)])
def test_hrnetpn_fused_section_against_the_oracle(width, size, B, tmp_path):
    """r05: the HRNetPN model's loss section as ONE autograd node (hip_ops.stage2_section_pn -- the C ABI's "absent
    second encoder": head 2 over the cloud features, rows of modality 2 gathered from the depth map) at small sizes,
    every output and gradient re-evaluated by oracle/check_step.py:check_section_pn (reference data flow:
    networks/build_backbone.py:457-514, learning/contrast_trainer.py:894-1039).  The BASELINE-size form of the same check
    is tests/test_whole_step_gpu.py::config4_hrnetpn_w32."""
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.learning.engine import RecordingEngine
    from oracle.check_step import check_records
    dev = torch.device('cuda:0')
    args = bench.make_args(B, 512, 2048, size, 'mpii', 'nccl', str(tmp_path), 3, arch='HRNetPN', width=width)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    eng = RecordingEngine('fp32')
    tr = ContrastTrainer(args, engine=eng)
    tr.device = dev
    try:
        model, contrast, opt, data = bench.build(args, tr, dev)
        assert model.defer_heads and model.defer_projection
        it = iter(data)
        eng.armed = False
        tr.train_step(next(it), model, contrast, opt, True)
        eng.armed = True
        out = tr.train_step(next(it), model, contrast, opt, True)
        torch.cuda.synchronize()
        assert [r['kind'] for r in eng.records] == ['section_pn']
        rep = check_records(eng.records)
        print(rep)
        assert abs(float(out['loss']) - float(eng.records[0]['total'])) <= 1e-4 * abs(float(out['loss']))
    finally:
        _lib.torch_glue().set_async_wgrad(False)
