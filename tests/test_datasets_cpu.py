"""Dataset tuple producers (reference datasets/dataset.py:306-617, datasets/mpii_utils.py:14-65).
(1) the arithmetic around the image decoding against vectors recorded from the reference's own functions
    (tests/golden/gen_golden.py:gen_dataset -> dataset_tuple.npz), bit-exact;
(2) a miniature NTU + MPII tree written to disk with PIL: the dataset, the source-balancing sampler, the loader and
    two trainer steps through ``main_contrast.main`` (oracle engine, CPU)."""
import json
import os
import pickle

import numpy as np
import pytest
from conftest import free_port
import torch
from PIL import Image

from conftest import load_golden
from hcmoco_amd.pycontrast.datasets import ntu_mpii as N
from hcmoco_amd.pycontrast.datasets.util import DistributedSamplerWrapper, source_balancing_weights


def npz():
    return load_golden('dataset_tuple')


def test_mpii_records_match_reference(tmp_path):
    g = npz()
    os.makedirs(tmp_path / 'annot')
    (tmp_path / 'annot' / 'train.json').write_text(str(g['mpii_anno_json']))
    db = N.mpii_records(str(tmp_path), 'train')
    assert np.array_equal(np.stack([r['center'] for r in db]), g['mpii_center'].numpy())
    assert np.array_equal(np.stack([r['scale'] for r in db]), g['mpii_scale'].numpy())
    assert np.array_equal(np.stack([r['joints_3d'] for r in db]), g['mpii_joints'].numpy())
    assert np.array_equal(np.stack([r['joints_3d_vis'] for r in db]), g['mpii_joints_vis'].numpy())
    assert [os.path.relpath(r['image'], tmp_path) for r in db] == [str(s) for s in g['mpii_image']]


def test_joint_bookkeeping_matches_reference():
    g = npz()
    assert np.array_equal(N.kinect_to_mpii(g['kinect25'].numpy()), g['kinect2mpii'].numpy())
    j16 = g['joints16'].numpy()
    assert np.array_equal(N.normalize_joints(j16), g['norm_myway'].numpy())
    assert np.array_equal(N.flip_normalized_joints(N.normalize_joints(j16).copy()), g['norm_flipped'].numpy())
    vis = g['vis16'].numpy().astype(bool)
    assert float(N.scale_from_joints(j16.astype(np.float64), vis)) == float(g['scale_mpii'])
    assert float(N.scale_from_joints(j16.astype(np.float64), np.zeros(16, bool))) == float(g['scale_mpii_none']) == 80.0


def test_affine_matches_reference():
    g = npz()
    for k in range(3):
        cx, cy, sx, sy, r = [float(v) for v in g['affine%d_in' % k]]
        t = N.affine_from_center_scale(np.array([cx, cy]), np.array([sx, sy]), r, (256, 256))
        assert np.allclose(t, g['affine%d' % k].numpy(), rtol=0, atol=1e-12)
        assert np.allclose(N.affine_point(np.array([31.0, 77.0]), t), g['affine%d_pt' % k].numpy(), atol=1e-10)
    # the matrix maps the box centre to the image centre and keeps distances scaled by out / (200 * scale)
    t = N.affine_from_center_scale(np.array([100.0, 60.0]), np.array([1.0, 1.0]), 0.0, (256, 256))
    assert np.allclose(N.affine_point([100.0, 60.0], t), [128.0, 128.0])
    assert np.allclose(N.affine_point([200.0, 60.0], t), [128.0 + 100 * 256 / 200.0, 128.0])


def test_ntu_tuple_matches_reference():
    """Items 0-8 for an NTU frame from an injected decoded frame: normalised depth, mask, joint re-ordering,
    visibility (with the reference's column quirk), flipped normalised skeleton, crop-relative pixel joints, scale."""
    g = npz()
    rp = [int(v) for v in g['ntu_resize_param']]
    resize_param = (rp[0], rp[1], rp[2], rp[3], bool(g['ntu_need_flip']), rp[4], rp[5])
    out = N.ntu_tuple(g['ntu_rgbd_in'].clone(), 5, g['ntu_joints3d'], resize_param, g['ntu_dloc'].numpy(),
                      int(g['ntu_size']), random_flip=True)
    names = ['rgbd', 'index', 'norm_joints', 'joints3d', 'original_joints2d', 'joints_vis', 'true_depth', 'depth_mask', 'scale']
    for n, v in zip(names, out):
        want = g['ntu_out_' + n]
        if isinstance(want, torch.Tensor):
            got = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            assert got.dtype == want.dtype and torch.equal(got, want), n
        else:
            assert float(v) == float(want), n
    assert int(out[5].sum()) not in (0, 16)            # the fixture has visible and invisible joints
    # a frame that was NOT mirrored: the reference mirrors the normalised skeleton all the same under --random_flip
    # (it tests resize_param[-1] = original_w, dataset.py:589); mirrored by default, switchable
    noflip = resize_param[:4] + (False,) + resize_param[5:]
    out_nf = N.ntu_tuple(g['ntu_rgbd_in'].clone(), 5, g['ntu_joints3d'], noflip, g['ntu_dloc'].numpy(),
                         int(g['ntu_size']), random_flip=True)
    assert torch.equal(out_nf[2], g['ntu_noflip_out_norm_joints'])
    assert torch.equal(out_nf[4], g['ntu_noflip_out_original_joints2d'])
    assert torch.equal(out_nf[2], out[2])
    try:
        N.REFERENCE_FLIP_QUIRK = False
        intended = N.ntu_tuple(g['ntu_rgbd_in'].clone(), 5, g['ntu_joints3d'], noflip, g['ntu_dloc'].numpy(),
                               int(g['ntu_size']), random_flip=True)
        assert not torch.equal(intended[2], out[2])
    finally:
        N.REFERENCE_FLIP_QUIRK = True


def test_warp_affine_identity_shift_and_border():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 255, (20, 30, 3)).astype(np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(N.warp_affine(img, ident, (30, 20)), img)
    shift = np.array([[1.0, 0, 3], [0, 1.0, -2]])              # dst(x, y) = src(x - 3, y + 2)
    out = N.warp_affine(img, shift, (30, 20))
    assert np.array_equal(out[0:18, 3:30], img[2:20, 0:27]) and int(out[:, :3].sum()) == 0 and int(out[18:].sum()) == 0
    half = np.array([[0.5, 0, 0], [0, 0.5, 0]])                # down-scale by 2: samples every second pixel exactly
    assert np.array_equal(N.warp_affine(img, half, (15, 10)), img[::2, ::2])


def test_crop_params_stay_inside_and_respect_ratio():
    import random
    rng = random.Random(3)
    for _ in range(200):
        top, left, h, w = N.crop_params(1920, 1080, (0.08, 1.0), (1, 1), rng)
        assert 0 <= top <= 1080 - h and 0 <= left <= 1920 - w and h == w and h > 0
    top, left, h, w = N.crop_params(1920, 1080, (3.0, 4.0), (1, 1), rng)       # impossible area: central fallback
    assert (h, w) == (1080, 1080) and top == 0 and left == (1920 - 1080) // 2


def test_sampler_weights_and_distributed_wrapper():
    w = source_balancing_weights(3, 9)
    assert np.isclose(w[:3].sum(), w[3:].sum())                # both sources carry the same probability mass
    base = torch.utils.data.WeightedRandomSampler(w, 12)
    a = DistributedSamplerWrapper(base, num_replicas=2, rank=0, shuffle=False)
    b = DistributedSamplerWrapper(base, num_replicas=2, rank=1, shuffle=False)
    torch.manual_seed(5)
    ia = list(a)
    torch.manual_seed(5)
    ib = list(b)
    torch.manual_seed(5)
    drawn = list(base)
    assert len(ia) == len(ib) == 6 and sorted(ia + ib) == sorted(drawn)        # the two ranks split ONE draw


def _write_tree(root, n_ntu=5, n_mpii=3, seed=0):
    """A miniature copy of the on-disk layout the reference expects (dataset.py:85-93, :165-172, :330-381)."""
    rng = np.random.RandomState(seed)
    rel = []
    for k in range(n_ntu):
        name = 'nturgb+d_rgb_warped_correction/S001C001P001R001A001/WRGB-%08d.jpg' % (k + 1)
        rel.append(name)
        rgb = (rng.rand(108, 192, 3) * 255).astype(np.uint8)
        depth = np.zeros((108, 192), np.uint16)
        depth[20:90, 60:130] = (2000 + rng.rand(70, 70) * 1500).astype(np.uint16)
        skel = {'joints': [{'3d_loc': (rng.randn(25, 3)).tolist(),
                            'd_loc': (np.array([60, 20]) + rng.rand(25, 2) * np.array([70, 70])).tolist()}]}
        for path, writer in ((name, lambda p: Image.fromarray(rgb).save(p, quality=95)),
                             (N.NTUMPIIContrastDataset._sibling(name, 'HumanRGBD/NTURGBD/nturgb+d_depth_masked', 'MDepth', 'png'),
                              lambda p: Image.fromarray(depth).save(p)),
                             (N.NTUMPIIContrastDataset._skeleton_name(name), lambda p: pickle.dump(skel, open(p, 'wb')))):
            full = os.path.join(root, path)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            writer(full)
    flist = os.path.join(root, 'flist.txt')
    open(flist, 'w').write('\n'.join(rel) + '\n')
    mpii = os.path.join(root, 'mpii')
    os.makedirs(os.path.join(mpii, 'annot'))
    os.makedirs(os.path.join(mpii, 'images'))
    anno = []
    for k in range(n_mpii):
        Image.fromarray((rng.rand(120, 160, 3) * 255).astype(np.uint8)).save(os.path.join(mpii, 'images', 'm%d.jpg' % k))
        anno.append({'image': 'm%d.jpg' % k, 'center': [80.0, 55.0], 'scale': 0.5,
                     'joints': (np.array([30, 10]) + rng.rand(16, 2) * 100).tolist(), 'joints_vis': [1] * 14 + [0, 1]})
    json.dump(anno, open(os.path.join(mpii, 'annot', 'train.json'), 'w'))
    return flist, mpii


def test_dataset_yields_the_positional_tuple(tmp_path):
    flist, mpii = _write_tree(str(tmp_path))
    ds = N.NTUMPIIContrastDataset(str(tmp_path), flist, mpii, 'train', size=64, random_flip=True, random_resized_crop=True)
    assert len(ds) == 8 and len(ds.db) == 3
    for index in (0, 2, 3, 7):
        t = ds[index]
        assert len(t) == 9
        rgbd, idx, norm, j3, orig, vis, true_depth, mask, scale = t
        assert rgbd.shape == (6, 64, 64) and rgbd.dtype == torch.float32 and idx == index
        assert norm.shape == (16, 2) and norm.dtype == torch.float32 and float(norm.abs().max()) <= 1.0 + 1e-6
        assert j3.shape == (25, 3) and orig.shape == (16, 2) and vis.shape == (16,) and vis.dtype == torch.int32
        assert mask.shape == (64, 64) and mask.dtype == torch.float32 and float(scale) > 0
        assert bool((orig[vis == 0] == 0).all()) and bool((norm[vis == 0] == 0).all())
        if index < 3:              # MPII: RGB only
            assert true_depth == 0 and float(mask.sum()) == 0 and float(rgbd[3:].abs().sum()) == 0 and int(vis[14]) == 0
        else:                      # NTU: masked, mean-centred depth replicated on three channels
            assert true_depth == 1 and float(mask.sum()) > 0
            d = rgbd[3]
            assert torch.equal(rgbd[4], d) and torch.equal(rgbd[5], d) and bool((d[mask == 0] == 0).all())
            assert abs(float(d[mask > 0].mean())) < 1e-3
    # no augmentation: the NTU frame is decoded as is (normalised RGB, depth in metres minus its mean)
    plain = N.NTUMPIIContrastDataset(str(tmp_path), flist, None, size=64)
    rgbd = plain[0][0]
    img = np.array(Image.open(plain.image_list[0]).convert('RGB'), dtype=np.float32) / 255.0
    want = (torch.from_numpy(img) - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])
    assert torch.allclose(rgbd[:3], want.permute(2, 0, 1).float(), atol=1e-6)


def test_image_dataset_drives_the_training_loop(tmp_path, monkeypatch):
    """``main_contrast.py --dataset NTUMPII`` (no --synthetic): files -> loader -> positional tuple -> two stage-2
    steps of the real trainer (oracle loss engine on CPU); MPII samples enter with use_depth = 0."""
    from hcmoco_amd.pycontrast import main_contrast
    from oracle.oracle_engine import OracleLossEngine
    flist, mpii = _write_tree(str(tmp_path / 'data'), n_ntu=6, n_mpii=4)
    monkeypatch.setenv('MASTER_PORT', str(free_port()))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SLURM_PROCID'):
        monkeypatch.delenv(k, raising=False)
    argv = ['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18', '--in_channel_list',
            '3,3', '--batch_size', '4', '--nce_k', '8', '--world-size', '1', '--dist-backend', 'gloo', '--dataset', 'NTUMPII',
            '--data_folder', str(tmp_path / 'data'), '--train_file_list', flist, '--mpii_root', mpii, '--image_size', '64',
            '--num_workers', '0', '--epochs', '1', '--print_freq', '1', '--save_freq', '1', '--model_path', str(tmp_path),
            '--tb_path', str(tmp_path), '--seed', '1', '--learning_rate', '0.01', '--linear_feat_map', '1',
            '--modality_missing', '1', '--pri3d_num_samples_per_image', '8', '--skeleton_meta_name', 'mpii', '--random_flip', '1']
    try:
        outs, trainer, model, contrast = main_contrast.main(argv, engine=OracleLossEngine())
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    assert len(outs) == 4 and outs[0] == outs[0]
    assert contrast.memory_1.shape[0] == 10              # the bank is sized by len(train_dataset) (main_contrast.py:49)


# ---------------------------------------------------------------------------------- NTU + COCO (13 joints)
def test_coco_records_and_reductions_match_reference(tmp_path):
    """person_keypoints json -> records through the reference's own loader (dataset.py:698-770, with the five
    pycocotools calls replaced by a json reader) vs ``coco_records`` reading the json directly: crowd, non-person,
    zero-area and keypoint-less annotations dropped, boxes clipped, centre/scale; COCOReduce / KinectReduce."""
    g = npz()
    os.makedirs(tmp_path / 'annotations')
    (tmp_path / 'annotations' / 'person_keypoints_train2014.json').write_text(str(g['coco_anno_json']))
    db = N.coco_records(str(tmp_path), 'train2014')
    assert [os.path.relpath(r['image'], tmp_path) for r in db] == [str(s) for s in g['coco_image']]
    assert np.array_equal(np.stack([r['center'] for r in db]), g['coco_center'].numpy())
    assert np.array_equal(np.stack([r['scale'] for r in db]), g['coco_scale'].numpy())
    assert np.array_equal(np.stack([r['joints_3d'] for r in db]), g['coco_joints'].numpy())
    assert np.array_equal(np.stack([r['joints_3d_vis'] for r in db]), g['coco_joints_vis'].numpy())
    ds = N.NTUCOCOContrastDataset.__new__(N.NTUCOCOContrastDataset)
    n, o, v = ds._reduce(g['coco_in_norm'].numpy(), g['coco_in_orig'].numpy(), g['coco_in_vis'].numpy())
    assert np.array_equal(n, g['coco_red_norm'].numpy()) and np.array_equal(o, g['coco_red_orig'].numpy())
    assert np.array_equal(v, g['coco_red_vis'].numpy())
    k = g['kinect25'].numpy()[N.KINECT_TO_REDUCED].reshape(13, 2)
    assert np.array_equal(k, g['kinect_reduce'].numpy())


def test_ntucoco_tuple_matches_reference():
    g = npz()
    rp = [int(v) for v in g['ntu_resize_param']]
    resize_param = (rp[0], rp[1], rp[2], rp[3], bool(g['ntu_need_flip']), rp[4], rp[5])
    out = N.ntu_tuple(g['ntu_rgbd_in'].clone(), 2, g['ntu_joints3d'], resize_param, g['ntu_dloc'].numpy(),
                      int(g['ntu_size']), random_flip=False, select=N.KINECT_TO_REDUCED, flip_pairs=N.COCO_FLIP_PAIRS)
    names = ['rgbd', 'index', 'norm_joints', 'joints3d', 'original_joints2d', 'joints_vis', 'true_depth', 'depth_mask', 'scale']
    for n, v in zip(names, out):
        want = g['ntucoco_out_' + n]
        if isinstance(want, torch.Tensor):
            got = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            assert got.dtype == want.dtype and torch.equal(got, want), n
        else:
            assert float(v) == float(want), n
    assert out[2].shape == (13, 2)


def test_ntucoco_dataset_end_to_end(tmp_path, monkeypatch):
    """A miniature COCO + NTU tree -> 13-joint tuples -> two stage-2 steps with the coco_reduce skeleton."""
    from hcmoco_amd.pycontrast import main_contrast
    from oracle.oracle_engine import OracleLossEngine
    root = str(tmp_path / 'data')
    flist, _ = _write_tree(root, n_ntu=5, n_mpii=1)
    coco = os.path.join(root, 'coco')
    os.makedirs(os.path.join(coco, 'annotations'))
    os.makedirs(os.path.join(coco, 'images', 'train2014'))
    rng = np.random.RandomState(1)
    data = {'categories': [{'id': 1, 'name': 'person'}], 'images': [], 'annotations': []}
    for k in range(3):
        Image.fromarray((rng.rand(120, 160, 3) * 255).astype(np.uint8)).save(
            os.path.join(coco, 'images', 'train2014', 'COCO_train2014_%012d.jpg' % (k + 1)))
        data['images'].append({'id': k + 1, 'width': 160, 'height': 120})
        kp = []
        for q in range(17):
            kp += [float(20 + rng.rand() * 100), float(10 + rng.rand() * 90), 2 if q != 3 else 0]
        data['annotations'].append({'id': k, 'image_id': k + 1, 'category_id': 1, 'iscrowd': 0, 'area': 5000.0,
                                    'bbox': [15.0, 5.0, 120.0, 100.0], 'keypoints': kp})
    json.dump(data, open(os.path.join(coco, 'annotations', 'person_keypoints_train2014.json'), 'w'))
    ds = N.NTUCOCOContrastDataset(root, flist, coco, 'train2014', size=64, random_resized_crop=True)
    assert len(ds) == 8 and len(ds.db) == 3
    t = ds[0]
    assert t[2].shape == (13, 2) and t[4].shape == (13, 2) and t[5].shape == (13,) and t[6] == 0
    assert ds[5][2].shape == (13, 2) and ds[5][6] == 1
    flip = N.NTUCOCOContrastDataset(root, flist, coco, 'train2014', size=64, random_flip=True, random_resized_crop=True)
    import random
    random.seed(0)
    with pytest.raises(IndexError, match='flip pairs'):          # the reference fails the same way (:820-826, :936-937)
        for _ in range(20):
            flip[6]
    monkeypatch.setenv('MASTER_PORT', str(free_port()))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SLURM_PROCID'):
        monkeypatch.delenv(k, raising=False)
    argv = ['--method', 'CMCJointsPri3DRGBD2S', '--modal', 'RGBD2S', '--arch', 'HRNet', '--width', '18', '--in_channel_list',
            '3,3', '--batch_size', '4', '--nce_k', '6', '--world-size', '1', '--dist-backend', 'gloo', '--dataset', 'NTUCOCO',
            '--data_folder', root, '--train_file_list', flist, '--coco_root', coco, '--image_size', '64', '--num_workers', '0',
            '--epochs', '1', '--print_freq', '1', '--save_freq', '1', '--model_path', str(tmp_path), '--tb_path', str(tmp_path),
            '--seed', '1', '--learning_rate', '0.01', '--linear_feat_map', '1', '--modality_missing', '1',
            '--pri3d_num_samples_per_image', '8', '--skeleton_meta_name', 'coco_reduce']
    try:
        outs, trainer, model, contrast = main_contrast.main(argv, engine=OracleLossEngine())
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    assert len(outs) == 4 and outs[0] == outs[0] and contrast.memory_1.shape[0] == 8


# ---------------------------------------------------------------------------------- NTU segmentation frames (HRNetPN)
@pytest.mark.parametrize('tag,index,mask_depth,mask_rgb', [('plain', 1, False, False), ('parsing', 4, False, False),
                                                           ('nodepth', 4, True, False), ('norgb', 3, False, True)])
def test_ntuseg_tuple_matches_reference(tag, index, mask_depth, mask_rgb, tmp_path):
    """The 16-item tuple of NTURGBDSegJoint.__getitem__ (dataset.py:1037-1116) from an injected decoded frame: label
    mapping of the parsing frames (255 elsewhere), true_label / true_depth / true_rgb with the two masking modes,
    grid_xy, frame size and the subtracted mean depth."""
    g = npz()
    rp = [int(v) for v in g['ntu_resize_param']]
    resize_param = (rp[0], rp[1], rp[2], rp[3], bool(g['ntu_need_flip']), rp[4], rp[5])
    size = int(g['ntu_size'])
    ds = N.NTUSegContrastDataset.__new__(N.NTUSegContrastDataset)
    ds.size, ds.random_flip, ds.random_resized_crop, ds.only_seg, ds.split = (size, size), False, True, False, 3
    ds.mask_seg_depth, ds.mask_seg_rgb, ds.label_mapper = mask_depth, mask_rgb, N.seg_label_mapper()
    t = N.ntu_tuple(g['ntu_rgbd_in'].clone(), index, g['ntu_joints3d'], resize_param, g['ntu_dloc'].numpy(), size, False,
                    empty_ok=True, with_mean=True)
    label_img = Image.fromarray(g['seg_label_png'].numpy().astype(np.uint8))
    rgbd, true_depth, mask, label, true_label, true_rgb, grid_xy, oh, ow = ds.seg_items(index, t[0], t[7], resize_param, label_img)
    got = (rgbd, index, t[2], t[3], t[4], t[5], true_depth, mask.float(), t[8], label, true_label, true_rgb, grid_xy, oh, ow,
           float(t[9]))
    names = ['rgbd', 'index', 'norm_joints', 'joints3d', 'original_joints2d', 'joints_vis', 'true_depth', 'depth_mask', 'scale',
             'label', 'true_label', 'true_rgb', 'grid_xy', 'original_h', 'original_w', 'mean']
    for n, v in zip(names, got):
        want = g['seg_%s_%s' % (tag, n)]
        if isinstance(want, torch.Tensor):
            v = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
            assert v.dtype == want.dtype and torch.equal(v, want), (tag, n)
        else:
            assert float(v) == float(want), (tag, n)
    assert (true_label == 1) == (index >= 3) and int(label.max()) <= (24 if index >= 3 else 255)


def test_ntuseg_dataset_feeds_the_hrnetpn_tuple(tmp_path):
    """Files on disk -> the 16-item tuple HRNetPN's trainer path consumes (items 12-15 = grid_xy, frame size, mean)."""
    root = str(tmp_path / 'data')
    flist, _ = _write_tree(root, n_ntu=3, n_mpii=1)
    seg = os.path.join(root, 'seg')
    rng = np.random.RandomState(4)
    lines = []
    for k in range(2):
        stem = 'S001C002P003R001A0%02dF%03d' % (k + 1, k + 5)
        lines.append('images/%s.jpg' % stem)
        skel_path = N.NTUSegContrastDataset._seg_skeleton(lines[-1], os.path.join(root, 'skel'))
        depth = np.zeros((108, 192), np.uint16)
        depth[30:80, 70:120] = 2500
        for path, writer in ((os.path.join(seg, 'images', stem + '.jpg'), lambda p: Image.fromarray((rng.rand(108, 192, 3) * 255).astype(np.uint8)).save(p)),
                             (os.path.join(seg, 'depth', 'MDepth-' + stem + '.png'), lambda p: Image.fromarray(depth).save(p)),
                             (os.path.join(seg, 'png_annotation_v2', stem + '.png'),
                              lambda p: Image.fromarray(rng.choice(np.array(N.SEG_ORIGINAL_LABELS), size=(108, 192)).astype(np.uint8)).save(p)),
                             (skel_path, lambda p: pickle.dump({'joints': [{'3d_loc': rng.randn(25, 3).tolist(),
                                                                            'd_loc': (np.array([70, 30]) + rng.rand(25, 2) * 50).tolist()}]},
                                                               open(p, 'wb')))):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            writer(path)
    seg_list = os.path.join(seg, 'train.txt')
    open(seg_list, 'w').write('\n'.join(reversed(lines)) + '\n')
    ds = N.NTUSegContrastDataset(root, flist, seg, seg_list, size=64, random_resized_crop=True, mask_seg_rgb=True,
                                 skeleton_root=os.path.join(root, 'skel'))
    assert len(ds) == 5 and ds.split == 3 and ds.skeleton_list[3].endswith('S001C002P003R001A001/Skeleton-00000005.pkl')
    plain, parsing = ds[0], ds[4]
    assert len(plain) == len(parsing) == 16
    assert plain[10] == 0 and plain[11] == 1 and int(plain[9].min()) == 255
    assert parsing[10] == 1 and parsing[11] == 0 and int(parsing[9].max()) <= 24 and float(parsing[0][:3].abs().sum()) == 0
    gxy = parsing[12]
    assert gxy.shape == (64, 64, 2) and gxy.dtype == torch.int32 and parsing[13] == 108 and parsing[14] == 192
    assert bool((gxy[..., 0][1:] >= gxy[..., 0][:-1]).all()) and bool((gxy[..., 1][:, 1:] >= gxy[..., 1][:, :-1]).all())
    assert isinstance(parsing[15], float) and parsing[15] > 0
    batch = torch.utils.data.default_collate([ds[3], ds[4]])
    assert batch[12].shape == (2, 64, 64, 2) and batch[13].tolist() == [108, 108] and batch[15].dtype == torch.float64
