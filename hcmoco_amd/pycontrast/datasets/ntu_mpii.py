"""NTU RGB-D + MPII / COCO tuple producers (``--dataset NTUMPII | NTUCOCO --modal RGBD2S``): the positional batch tuple the
trainer indexes from image files.

Reference: /root/reference/pycontrast/datasets/dataset.py:65-250 (NTU frame: RGB jpg, masked 16-bit depth png,
parsed-skeleton pickle), :306-381 (MPII annotation records), :474-617 (the GCN variant whose ``__getitem__``
yields the tuple) and datasets/mpii_utils.py:14-65 (centre/scale affine).  Index space as in the reference:
``[0, len(mpii))`` are MPII images (RGB only: zero depth, ``use_depth = 0``), the rest NTU frames.

The reference decodes and warps with cv2 / torchvision, neither of which is in this image; this module uses
PIL + numpy:
  * NTU crop / resize: ``PIL.Image.crop`` + ``resize`` -- the two calls ``torchvision.transforms.functional.
    resized_crop`` makes on PIL images (bilinear for RGB, nearest for depth);
  * the random crop rectangle: torchvision's ``RandomResizedCrop.get_params`` restated (``crop_params``);
  * MPII: ``cv2.warpAffine(INTER_LINEAR)`` restated as an inverse-mapped bilinear sample with a zero border
    (``warp_affine``) -- float interpolation, not cv2's 5-bit fixed point: pixel values can differ in the last
    bits of uint8.
Everything AROUND the decoding -- annotation records, joint re-ordering / normalisation / flipping,
visibility, depth normalisation, the `scale` heuristic, the affine matrix -- is pinned against the
reference's own functions (tests/golden/dataset_tuple.npz, tests/test_datasets_cpu.py), including the
reference's quirk of testing ``joints2d[:, 1] < j + w`` where it means column 0 (dataset.py:591-592).
"""
import json
import math
import os
import pickle
import random

import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)
KINECT_TO_MPII = [14, 13, 12, 16, 17, 18, 0, 1, 2, 3, 6, 5, 4, 8, 9, 10]          # dataset.py:325-328
MPII_FLIP_PAIRS = [[0, 5], [1, 4], [2, 3], [10, 15], [11, 14], [12, 13]]           # dataset.py:480


# --------------------------------------------------------------------------- pure arithmetic (pinned)
def mpii_records(root, image_set='train', num_joints=16):
    """Annotation json -> list of dicts (centre / scale padded, 1-based -> 0-based; dataset.py:330-381)."""
    with open(os.path.join(root, 'annot', image_set + '.json')) as f:
        anno = json.load(f)
    out = []
    for a in anno:
        c = np.array(a['center'], dtype=np.float64)
        s = np.array([a['scale'], a['scale']], dtype=np.float64)
        if c[0] != -1:                          # keep limbs inside the crop
            c[1] = c[1] + 15 * s[1]
            s = s * 1.25
        c = c - 1
        joints = np.zeros((num_joints, 3), dtype=np.float64)
        vis = np.zeros((num_joints, 3), dtype=np.float64)
        if image_set != 'test':
            j = np.array(a['joints'], dtype=np.float64)
            assert len(j) == num_joints, 'joint num diff: {} vs {}'.format(len(j), num_joints)
            joints[:, 0:2] = j[:, 0:2] - 1
            v = np.array(a['joints_vis'])
            vis[:, 0] = v
            vis[:, 1] = v
        out.append({'image': os.path.join(root, 'images', a['image']), 'center': c, 'scale': s,
                    'joints_3d': joints, 'joints_3d_vis': vis})
    return out


def kinect_to_mpii(joints25):
    return np.asarray(joints25)[KINECT_TO_MPII].reshape(16, 2)


def normalize_joints(joints2d, root_index=6):
    """root-centred, (x, y) -> (y, x), scaled to max-abs 1 (dataset.py:482-488)."""
    j = np.array(joints2d, copy=True)
    j = j - j[root_index, :]
    j = j[:, ::-1]
    s = max(j.max(), np.abs(j.min()))
    return j / s


def flip_normalized_joints(norm_joints, pairs=MPII_FLIP_PAIRS):
    """mirror the second coordinate and swap left/right joints, in place like the reference (:494-500)."""
    norm_joints[:, 1] = -norm_joints[:, 1]
    tmp = norm_joints.copy()
    for i, j in pairs:
        norm_joints[i, :] = tmp[j, :]
        norm_joints[j, :] = tmp[i, :]
    return norm_joints


def scale_from_joints(joint2d, joint_vis):
    """largest distance between two visible joints; 80 when there is none (dataset.py:457-472)."""
    n = joint2d.shape[0]
    d = joint2d.reshape(n, 1, 2) - joint2d.reshape(1, n, 2)
    d = np.sqrt((d ** 2).sum(-1))
    d[~joint_vis, :] = -1
    d[:, ~joint_vis] = -1
    m = d.max()
    return 80 if (m == -1 or m == 0) else m


def _rot(point, rad):
    sn, cs = np.sin(rad), np.cos(rad)
    return [point[0] * cs - point[1] * sn, point[0] * sn + point[1] * cs]


def _third(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def affine_from_center_scale(center, scale, rot, output_size):
    """2x3 matrix mapping the (centre, 200*scale box, rotation) source frame onto the output image
    (mpii_utils.py:28-60; cv2.getAffineTransform = the exact solve of three point pairs)."""
    scale_tmp = np.asarray(scale, dtype=np.float64) * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    src_dir = _rot([0, src_w * -0.5], np.pi * rot / 180)
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :] = _third(src[0, :], src[1, :])
    dst[2, :] = _third(dst[0, :], dst[1, :])
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i] = [src[i, 0], src[i, 1], 1, 0, 0, 0]
        a[2 * i + 1] = [0, 0, 0, src[i, 0], src[i, 1], 1]
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def affine_point(pt, t):
    return np.dot(t, np.array([pt[0], pt[1], 1.0]))[:2]


def warp_affine(img, t, out_size):
    """``cv2.warpAffine(img, t, out_size, flags=INTER_LINEAR)`` for an HxWxC uint8/float array: every output
    pixel samples the source at t^-1 (x, y) bilinearly; outside the image is 0 (BORDER_CONSTANT)."""
    w_out, h_out = out_size
    m = np.vstack([t, [0, 0, 1]])
    inv = np.linalg.inv(m)[:2]
    ys, xs = np.mgrid[0:h_out, 0:w_out].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    h, w = img.shape[:2]
    src = img.astype(np.float64)

    def at(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return v * ok[..., None]
    out = ((1 - fy) * ((1 - fx) * at(y0, x0) + fx * at(y0, x0 + 1)) + fy * ((1 - fx) * at(y0 + 1, x0) + fx * at(y0 + 1, x0 + 1)))
    return np.clip(np.rint(out), 0, 255).astype(np.uint8) if img.dtype == np.uint8 else out.astype(img.dtype)


def crop_params(width, height, scale, ratio, rng=random):
    """torchvision.transforms.RandomResizedCrop.get_params restated: ten tries of (area fraction ~ U(scale),
    log-aspect ~ U(log ratio)), then the central crop at the clamped aspect.  Returns (top, left, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return rng.randint(0, height - h), rng.randint(0, width - w), h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


# The reference decides whether to mirror an NTU frame's NORMALISED skeleton by testing ``resize_param[-1]``
# (dataset.py:589, :927, :1050) -- that is ``original_w``, always truthy -- where it means ``need_flip``
# (resize_param[4]): under --random_flip every NTU skeleton is mirrored, whether or not the frame was.  Mirrored
# here for parity (golden: ntu_noflip_out_norm_joints); set to False for the evidently intended behaviour.
REFERENCE_FLIP_QUIRK = True


def skeleton_is_mirrored(resize_param):
    return bool(resize_param[-1]) if REFERENCE_FLIP_QUIRK else bool(resize_param[4])


def ntu_tuple(rgbd, index, joints3d, resize_param, d_loc, size, random_flip, select=KINECT_TO_MPII,
              flip_pairs=MPII_FLIP_PAIRS, empty_ok=False, with_mean=False):
    """Items 0-8 of the tuple for an NTU frame, from the decoded (cropped, resized, flipped, normalised) frame
    ``rgbd`` [6, size, size], the 25 Kinect depth-image joints ``d_loc`` and the crop rectangle
    ``resize_param = (i, j, h, w, need_flip, original_h, original_w)`` (dataset.py:578-617; :924-955 for the
    13-joint COCO skeleton: ``select`` = KinectReduce, ``flip_pairs`` = the COCO pairs)."""
    joints2d = np.array(d_loc, dtype=np.float32)[select].reshape(len(select), 2)
    i, j, h, w, need_flip = resize_param[:5]
    norm = normalize_joints(joints2d)
    if random_flip and skeleton_is_mirrored(resize_param):
        norm = flip_normalized_joints(norm, flip_pairs)
    # the reference compares column 1 against j + w in the last test (it means column 0); mirrored for parity
    vis = np.logical_and(np.logical_and(joints2d[:, 1] > i, joints2d[:, 1] < i + h),
                         np.logical_and(joints2d[:, 0] > j, joints2d[:, 1] < j + w))
    original = joints2d[:, ::-1].copy()
    original[:, 0] = (original[:, 0] - i) / h * size
    original[:, 1] = (original[:, 1] - j) / w * size
    depth = rgbd[3]
    mask = depth > 0
    mean = 0.0 if (empty_ok and mask.sum() == 0) else depth.sum() / mask.sum()     # (:1068-1071 guards the empty frame)
    centred = depth - mean
    centred[~mask] = 0
    rgbd = rgbd.clone()
    rgbd[3:] = centred.unsqueeze(0)
    original[np.logical_not(vis), :] = 0
    norm[np.logical_not(vis), :] = 0
    scale = scale_from_joints(original, vis)
    out = (rgbd, index, torch.from_numpy(norm.copy().astype(np.float32)), joints3d, torch.from_numpy(original.copy()),
           torch.from_numpy(vis.astype(np.int32).copy()), 1, mask.float(), scale)
    return out + (mean,) if with_mean else out


def _to_tensor_normalised(img_uint8_hwc):
    x = torch.from_numpy(np.array(img_uint8_hwc, dtype=np.float32))
    x /= 255.0
    x -= torch.from_numpy(IMAGENET_MEAN)
    x /= torch.from_numpy(IMAGENET_STD)
    return x.permute(2, 0, 1)


# --------------------------------------------------------------------------- the dataset
class NTUMPIIContrastDataset(torch.utils.data.Dataset):
    """``modal2Dataset['NTUMPIIRGBD2S']`` (= NTUMPIIRGBD3D2DSkeletonGCN): MPII images first, then NTU frames."""
    KINECT_SELECT = KINECT_TO_MPII       # Kinect joints that make up this skeleton, in its order
    FLIP_PAIRS = MPII_FLIP_PAIRS         # left/right pairs of the 2-D pose source's own joint order

    def _records(self, root, image_set):
        return mpii_records(root, image_set)

    def _reduce(self, norm, original, vis):
        return norm, original, vis

    def __init__(self, ntu_root, ntu_file_list, mpii_root, mpii_image_set='train', size=256, random_flip=False,
                 random_resized_crop=False):
        self.root = ntu_root
        self.file_list = [f.strip() for f in open(ntu_file_list)] if ntu_file_list else []
        self.size = (size, size)
        self.random_flip, self.random_resized_crop = random_flip, random_resized_crop
        self.image_list = [os.path.join(ntu_root, f) for f in self.file_list]
        self.depth_list = [os.path.join(ntu_root, self._sibling(f, 'HumanRGBD/NTURGBD/nturgb+d_depth_masked', 'MDepth', 'png'))
                           for f in self.file_list]
        self.skeleton_list = [os.path.join(ntu_root, self._skeleton_name(f)) for f in self.file_list]
        self.db = self._records(mpii_root, mpii_image_set) if mpii_root else []
        self.num_joints = 25

    @staticmethod
    def _sibling(f, prefix, tag, ext):
        return f.replace('nturgb+d_rgb_warped_correction', prefix).replace('WRGB', tag).replace('jpg', ext)

    @classmethod
    def _skeleton_name(cls, f):
        f = cls._sibling(f, 'HumanRGBD/NTURGBD/nturgb+d_parsed_skeleton', 'Skeleton', 'pkl')
        num = int(f[-12:-4])                     # skeleton frames are numbered from 0, images from 1 (:165-172)
        return f[:-12] + str(num - 1).zfill(8) + f[-4:]

    def __len__(self):
        return len(self.db) + len(self.image_list)

    # ---- MPII image -> items 0, 2, 4, 5 (dataset.py:502-562)
    def _mpii(self, k):
        rec = self.db[k]
        img = np.array(Image.open(rec['image']).convert('RGB'))
        joints, jvis = rec['joints_3d'].copy(), rec['joints_3d_vis']
        c, s, r = rec['center'], rec['scale'], 0
        if self.random_resized_crop:
            s = s * np.clip(np.random.randn() * 0.25 + 1, 0.75, 1.25)
            r = np.clip(np.random.randn() * 30, -60, 60) if random.random() < 0.6 else 0
        t = affine_from_center_scale(c, s, r, self.size)
        img = warp_affine(img, t, self.size)
        original = joints[:, :2].copy()
        if self.random_resized_crop:             # (the reference leaves the joints un-warped otherwise)
            for q in range(jvis.shape[0]):
                if jvis[q, 0] > 0.0:
                    original[q, 0:2] = affine_point(joints[q, 0:2], t)
        norm = normalize_joints(joints[:, :2])
        original = original[:, ::-1]
        if self.random_flip and random.random() <= 0.5:
            img = np.ascontiguousarray(img[:, ::-1, :])
            norm = flip_normalized_joints(norm, self.FLIP_PAIRS)
            original[:, 1] = self.size[1] - original[:, 1]
        x = _to_tensor_normalised(img)
        vis = np.logical_and(np.logical_and(np.logical_and(original[:, 0] >= 0, original[:, 0] < self.size[0]),
                                            np.logical_and(original[:, 1] >= 0, original[:, 1] < self.size[0])), jvis[:, 0])
        return torch.cat([x, torch.zeros_like(x)], 0), norm, original, vis

    # ---- NTU frame -> decoded rgbd + crop rectangle (dataset.py:175-250)
    def _ntu_frame(self, k):
        img = Image.open(self.image_list[k]).convert('RGB')
        depth = Image.open(self.depth_list[k])
        original_h, original_w = img.size[1], img.size[0]
        with open(self.skeleton_list[k], 'rb') as f:
            skel = pickle.load(f)
        body = skel['joints'][0]
        j3 = np.array(list(body['3d_loc']), dtype=np.float32)
        joints3d = torch.from_numpy(j3 - j3[0])
        if self.random_resized_crop:
            j2 = np.array(list(body['d_loc']))
            assert not np.any(np.isnan(j2)), self.skeleton_list[k]
            cx = random.randrange(int(j2[:, 1].min()), int(j2[:, 1].max()))
            cy = random.randrange(int(j2[:, 0].min()), int(j2[:, 0].max()))
            _, _, h, w = crop_params(img.size[0], img.size[1], (0.08, 1.2), (1, 1))
            i, j = int(cx - h / 2.0), int(cy - w / 2.0)
            box = (j, i, j + w, i + h)
            img = img.crop(box).resize(self.size[::-1], Image.BILINEAR)
            depth = depth.crop(box).resize(self.size[::-1], Image.NEAREST)
        else:
            i, j, h, w = 0, 0, img.size[0], img.size[1]
        need_flip = random.random() >= 0.5
        if self.random_flip and need_flip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
            depth = depth.transpose(Image.FLIP_LEFT_RIGHT)
        x = _to_tensor_normalised(np.array(img))
        d = torch.from_numpy(np.array(depth).astype(np.float32) / 1000.0)
        rgbd = torch.cat([x, torch.stack([d, d, d], 0)], 0)
        return rgbd, joints3d, (i, j, h, w, need_flip, original_h, original_w), body['d_loc']

    def __getitem__(self, index):
        if index < len(self.db):
            rgbd, norm, original, vis = self._mpii(index)
            norm, original, vis = self._reduce(norm, original, vis)
            original[np.logical_not(vis), :] = 0
            norm[np.logical_not(vis), :] = 0
            return (rgbd, index, torch.from_numpy(norm.copy().astype(np.float32)), torch.zeros([self.num_joints, 3]),
                    torch.from_numpy(original.copy()), torch.from_numpy(vis.astype(np.int32).copy()), 0,
                    torch.zeros_like(rgbd[0]), scale_from_joints(original, vis))
        rgbd, joints3d, resize_param, d_loc = self._ntu_frame(index - len(self.db))
        if (self.random_flip and skeleton_is_mirrored(resize_param)
                and max(max(p) for p in self.FLIP_PAIRS) >= len(self.KINECT_SELECT)):
            # the reference applies the 17-joint COCO pairs to its 13-joint NTU skeleton here and dies with an
            # IndexError (dataset.py:820-826, :936-937); its scripts never pass --random_flip for this dataset
            raise IndexError('flip pairs of the 2-D pose source do not fit the %d-joint NTU skeleton '
                             '(same failure as the reference; run without --random_flip)' % len(self.KINECT_SELECT))
        return ntu_tuple(rgbd, index, joints3d, resize_param, d_loc, self.size[0], self.random_flip,
                         self.KINECT_SELECT, self.FLIP_PAIRS)


# --------------------------------------------------------------------------- NTU + COCO (13-joint skeleton)
COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]      # dataset.py:651-652
COCO_TO_REDUCED = [16, 14, 12, 11, 13, 15, 0, 10, 8, 6, 5, 7, 9]                                # COCOReduce, :889-903
KINECT_TO_REDUCED = [14, 13, 12, 16, 17, 18, 3, 6, 5, 4, 8, 9, 10]                              # KinectReduce, :905-907


def box_to_center_scale(x, y, w, h, aspect_ratio=1.0, pixel_std=200):
    """COCO box -> (centre, scale) of the crop (dataset.py:776-791)."""
    center = np.zeros((2), dtype=np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * 1.25
    return center, scale


def coco_records(root, image_set='train2014', num_joints=17):
    """person_keypoints_<set>.json -> one record per annotated person (dataset.py:698-770), read with ``json``
    instead of pycocotools: images in id order, their non-crowd annotations in file order, boxes clipped to the
    image, persons without any labelled keypoint dropped."""
    prefix = 'person_keypoints' if 'test' not in image_set else 'image_info'
    with open(os.path.join(root, 'annotations', prefix + '_' + image_set + '.json')) as f:
        data = json.load(f)
    person = [c['id'] for c in data['categories']][0]            # class index 1 = the first category
    by_image = {}
    for a in data.get('annotations', []):
        if not a.get('iscrowd', 0):
            by_image.setdefault(a['image_id'], []).append(a)
    out = []
    for im in data['images']:
        width, height = im['width'], im['height']
        for obj in by_image.get(im['id'], []):
            x, y, w, h = obj['bbox']
            x1, y1 = np.max((0, x)), np.max((0, y))
            x2 = np.min((width - 1, x1 + np.max((0, w - 1))))
            y2 = np.min((height - 1, y1 + np.max((0, h - 1))))
            if not (obj['area'] > 0 and x2 >= x1 and y2 >= y1):
                continue
            if obj['category_id'] != person or max(obj['keypoints']) == 0:
                continue
            joints = np.zeros((num_joints, 3), dtype=np.float64)
            vis = np.zeros((num_joints, 3), dtype=np.float64)
            for k in range(num_joints):
                joints[k, 0], joints[k, 1] = obj['keypoints'][3 * k], obj['keypoints'][3 * k + 1]
                v = min(obj['keypoints'][3 * k + 2], 1)
                vis[k, 0], vis[k, 1] = v, v
            center, scale = box_to_center_scale(x1, y1, x2 - x1, y2 - y1)
            name = '%012d.jpg' % im['id']
            if '2014' in image_set:
                name = 'COCO_%s_' % image_set + name
            folder = 'test2017' if 'test' in image_set else image_set
            out.append({'image': os.path.join(root, 'images', folder, name), 'center': center, 'scale': scale,
                        'joints_3d': joints, 'joints_3d_vis': vis})
    return out


class NTUCOCOContrastDataset(NTUMPIIContrastDataset):
    """``modal2Dataset['NTUCOCORGBD2S']`` (= NTUCOCORGBD3D2DSkeletonGCN, dataset.py:622-955): COCO persons first
    (17 keypoints reduced to the 13-joint ``coco_reduce`` skeleton after augmentation), then NTU frames."""
    KINECT_SELECT = KINECT_TO_REDUCED
    FLIP_PAIRS = COCO_FLIP_PAIRS

    def _records(self, root, image_set):
        return coco_records(root, image_set)

    def _reduce(self, norm, original, vis):
        return (norm[COCO_TO_REDUCED].reshape(13, 2), original[COCO_TO_REDUCED].reshape(13, 2), vis[COCO_TO_REDUCED])


# --------------------------------------------------------------------------- NTU + NTU-segmentation frames (HRNetPN)
SEG_ORIGINAL_LABELS = [0, 1, 2, 3, 6, 7, 8, 17, 18, 19, 25, 26, 27, 32, 33, 34, 38, 39, 43, 44, 46, 49, 50, 56, 58]


def seg_label_mapper():
    """raw annotation value -> 0..24 (dataset.py:1016-1019); values outside the list map to themselves."""
    m = np.arange(60)
    for i, l in enumerate(SEG_ORIGINAL_LABELS):
        m[l] = i
    return m


class NTUSegContrastDataset(NTUMPIIContrastDataset):
    """``modal2Dataset['NTUSegRGBD2S']`` (= NTURGBDSegJoint, dataset.py:957-1120): the NTU frames of the file list
    followed by the frames of the human-parsing subset, every sample an NTU-style frame; the tuple carries seven
    more items -- 9 ``label`` (uint8 map, 255 where there is no annotation), 10 ``true_label``, 11 ``true_rgb``,
    12 ``grid_xy`` (source pixel of every crop pixel, what HRNetPN back-projects with), 13/14 the frame size,
    15 the mean depth that was subtracted.  ``mask_seg_depth`` / ``mask_seg_rgb`` blank one modality of the
    parsing frames (the "versatility" settings)."""

    def __init__(self, ntu_root, ntu_file_list, seg_root, seg_image_set, size=256, random_flip=False,
                 random_resized_crop=False, only_seg=False, mask_seg_depth=False, mask_seg_rgb=False,
                 skeleton_root='./data/NTURGBD'):
        super().__init__(ntu_root, ntu_file_list, None, size=size, random_flip=random_flip,
                         random_resized_crop=random_resized_crop)
        assert not random_flip, 'the parsing labels are not flipped (dataset.py:1089)'
        self.only_seg, self.mask_seg_depth, self.mask_seg_rgb = only_seg, mask_seg_depth, mask_seg_rgb
        lines = sorted(l.strip() for l in open(seg_image_set))
        stem = lambda fn: fn.split('/')[1].split('.')[0]
        seg_images = [os.path.join(seg_root, l) for l in lines]
        seg_depth = [os.path.join(seg_root, 'depth', 'MDepth-' + stem(l) + '.png') for l in lines]
        self.seg_gt_list = [os.path.join(seg_root, 'png_annotation_v2', stem(l) + '.png') for l in lines]
        seg_skel = [self._seg_skeleton(l, skeleton_root) for l in lines]
        self.split = len(self.image_list)
        if only_seg:
            self.image_list, self.depth_list, self.skeleton_list = seg_images, seg_depth, seg_skel
        else:
            self.image_list = self.image_list + seg_images
            self.depth_list = self.depth_list + seg_depth
            self.skeleton_list = self.skeleton_list + seg_skel
        self.label_mapper = seg_label_mapper()

    @staticmethod
    def _seg_skeleton(fn, skeleton_root):
        import re
        m = re.match(r'.*S(\d{3})C(\d{3})P(\d{3})R(\d{3})A(\d{3})F(\d{3}).*', fn)
        setup, frame = int(m.group(1)), int(m.group(6))
        tag = fn.split('/')[-1][:-8]                    # strip 'Fnnn.ext'
        return os.path.join(skeleton_root, 'NTURGBD' if setup < 18 else 'NTURGBD120', 'nturgb+d_parsed_skeleton', tag,
                            'Skeleton-{:08d}.pkl'.format(frame))

    def __len__(self):
        return len(self.image_list)

    def _crop_nearest(self, pil, resize_param):
        i, j, h, w = resize_param[:4]           # unconditional, as in the reference (labels and grid are always cropped)
        return pil.crop((j, i, j + w, i + h)).resize(self.size[::-1], Image.NEAREST)

    def seg_items(self, index, rgbd, mask, resize_param, label_image=None):
        """items 6-7 (possibly blanked) and 9-14 for frame ``index`` (dataset.py:1082-1116)."""
        i, j, h, w, _, original_h, original_w = resize_param
        true_depth, true_rgb = 1, 1
        parsing = index >= self.split or self.only_seg
        if parsing:
            if label_image is None:
                label_image = Image.open(self.seg_gt_list[index if self.only_seg else index - self.split])
            label = torch.from_numpy(self.label_mapper[np.array(self._crop_nearest(label_image, resize_param)).astype(np.uint8)])
            true_label = 1
        else:
            label = torch.zeros_like(rgbd[0], dtype=torch.uint8) + 255
            true_label = 0
        if self.mask_seg_depth and index >= self.split and not self.only_seg:
            true_depth, mask = 0, torch.zeros_like(rgbd[0])
            rgbd = torch.cat([rgbd[:3], torch.zeros_like(rgbd[:3])], 0)
        if self.mask_seg_rgb and index >= self.split and not self.only_seg:
            true_rgb = 0
            rgbd = torch.cat([torch.zeros_like(rgbd[:3]), rgbd[3:]], 0)
        gx, gy = torch.meshgrid(torch.arange(original_h), torch.arange(original_w), indexing='ij')
        gx = self._crop_nearest(Image.fromarray(gx.numpy().astype(np.uint16)), resize_param)
        gy = self._crop_nearest(Image.fromarray(gy.numpy().astype(np.uint16)), resize_param)
        grid_xy = torch.from_numpy(np.stack([np.array(gx), np.array(gy)], -1).astype(np.int32))
        return rgbd, true_depth, mask, label, true_label, true_rgb, grid_xy, int(original_h), int(original_w)

    def __getitem__(self, index):
        rgbd, joints3d, resize_param, d_loc = self._ntu_frame(index)
        t = ntu_tuple(rgbd, index, joints3d, resize_param, d_loc, self.size[0], self.random_flip, self.KINECT_SELECT,
                      self.FLIP_PAIRS, empty_ok=True, with_mean=True)
        rgbd, true_depth, mask, label, true_label, true_rgb, grid_xy, oh, ow = self.seg_items(index, t[0], t[7], resize_param)
        return (rgbd, index, t[2], t[3], t[4], t[5], true_depth, mask.float(), t[8], label, true_label, true_rgb, grid_xy,
                oh, ow, float(t[9]))
