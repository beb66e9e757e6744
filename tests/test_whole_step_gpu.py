"""ONE whole training step per BASELINE.json configuration at the configuration's OWN size, checked against the
oracle.  The step is the product's (``ContrastTrainer.train_step``: encoder programs, side streams,
HIP loss kernels, SGD, bank update); a ``RecordingEngine`` keeps what the loss kernels were given and what they
returned (for the HRNet models ONE record of the fused loss section, heads included), and ``oracle/check_step.py``
re-evaluates all of it on the CPU: the head features f, idx[:,0]==index bit-exact, validity of the sampled pixels, six bank losses
and accuracies, all B gradient rows per modality, the momentum update (touched rows to 1e-6, the others bit-identical),
the nine feature-map meters and the gradients of the feature-map losses w.r.t. the HRNet branch maps, the 1x1
projection weights and the SemGCN output (reference data flow: merge_all_res + full-resolution projection,
networks/build_backbone.py:243-254, :290-300; losses learning/contrast_trainer.py:954-980).

Tolerances (fp32) = SURVEY 8d's parity gate: losses and feature-map meters 1e-5 relative, bank and feature-map gradients
1e-4 relative L2 (r06: were 2e-4 / 5e-4; measured on the bench configuration 3.8e-7 / 3.8e-6), bank update 1e-6 absolute.  bf16 (config 5): bank rows are compared in the rows' own rounding (update to 1 bf16 ulp), the
bf16 feature-map contractions to 1e-2 on the meters and 2e-2 relative L2 on the gradients against the fp32 oracle."""
import os
import sys
import tempfile

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (arch, width, batch, size, K, bank_dtype, fmap_dtype, skeleton)
    'config2_stage2_w18_K16384': ('HRNet', 18, 32, 256, 16384, 'fp32', 'fp32', 'coco17'),
    'config2_mpii16': ('HRNet', 18, 32, 256, 16384, 'fp32', 'fp32', 'mpii'),
    'config3_per_gpu_half_K65536': ('HRNet', 18, 32, 256, 65536, 'fp32', 'fp32', 'coco17'),
    'config4_hrnetpn_w32': ('HRNetPN', 32, 32, 256, 16384, 'fp32', 'fp32', 'coco17'),
    'config5_bf16_K131072': ('HRNet', 18, 32, 256, 131072, 'bf16', 'bf16', 'coco17'),
    # the reference's recipe of record (scripts/SecondStage/train_ntumpiirgbd2s_hrnet_w18.sh:8-46: global batch 224 on
    # 4 GPUs = 56 per GPU, K = 16384, MPII 16 joints; datasets/dataset.py:475 size=320): HRNet maps 80^2 40^2 20^2 10^2
    'recipe_320_b56_mpii16': ('HRNet', 18, 56, 320, 16384, 'fp32', 'fp32', 'mpii'),
    # HRNetPN at 320^2: pts2depth interpolates n = 102 400 pixels from 4096 points (build_backbone.py:448-455)
    'recipe_320_hrnetpn_w32': ('HRNetPN', 32, 32, 320, 16384, 'fp32', 'fp32', 'coco17'),
    # first stage (scripts/FirstStage/train_ntumpiirgbd2s_hrnet_w18.sh) at the same crop and per-GPU batch
    'recipe_320_b56_stage1': ('HRNet', 18, 56, 320, 16384, 'fp32', 'fp32', 'mpii', 1),
}


@pytest.mark.parametrize('name', list(CONFIGS))
def test_one_step_at_the_configs_own_size_against_the_oracle(name):
    import bench
    from hcmoco_amd import _lib
    from hcmoco_amd.pycontrast.learning.contrast_trainer import ContrastTrainer
    from hcmoco_amd.pycontrast.learning.engine import RecordingEngine
    from oracle.check_step import check_records
    arch, width, B, size, K, bank_dtype, fmap_dtype, skeleton = CONFIGS[name][:8]
    stage2 = len(CONFIGS[name]) == 8
    dev = torch.device('cuda:0')
    args = bench.make_args(B, K, 131072, size, skeleton, 'nccl', tempfile.mkdtemp(), 2, arch=arch, width=width,
                           bank_dtype=bank_dtype, fmap_dtype=fmap_dtype)
    args.rank, args.world_size, args.local_rank, args.gpu, args.channels_last = 0, 1, 0, 0, False
    eng = RecordingEngine(fmap_dtype)
    tr = ContrastTrainer(args, engine=eng)
    tr.device = dev
    try:
        model, contrast, opt, data = bench.build(args, tr, dev)
        it = iter(data)
        eng.armed = False
        tr.train_step(next(it), model, contrast, opt, stage2)    # quiet-Find step (one stream, nothing deferred)
        eng.armed = True
        out = tr.train_step(next(it), model, contrast, opt, stage2)   # the DEFAULT runtime: this is the step checked
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out['loss']))
        kinds = [r['kind'] for r in eng.records]
        if not stage2:                       # stage 1 (contrast_trainer.py:532-640): heads in the model, bank NCE only
            assert kinds == ['bank'], kinds
            bank = eng.records[0]
            assert bank['idx'].shape == (B, K + 1) and bank['x'][0].shape == (B, 128)
            rep = check_records(eng.records)
            print(name, rep)
            assert abs(float(out['loss']) - float(bank['total'])) <= 1e-4 * abs(float(out['loss']))
            return
        # the fused loss section: ONE record holds rows 1-9 -- since r05 for the HRNetPN model too (second modality:
        # cloud features + depth map, hip_ops.stage2_section_pn, checked by oracle/check_step.py:check_section_pn)
        assert kinds == (['section_pn'] if arch == 'HRNetPN' else ['section']), kinds
        fm = eng.records[0]
        assert fm['idx'].shape == (B, K + 1) and fm['f'].shape == (B, 384)
        total = float(fm['total'])
        assert fm['sample_ind'].shape == (B, 400)
        rep = check_records(eng.records)
        print(name, rep)
        assert abs(float(out['loss']) - total) <= 1e-4 * abs(float(out['loss']))
    finally:
        _lib.torch_glue().set_async_wgrad(False)
