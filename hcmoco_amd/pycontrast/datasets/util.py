"""``build_own_contrast_loader`` for image datasets (reference: /root/reference/pycontrast/datasets/util.py:456-585):
the source-balancing ``WeightedRandomSampler``, its distributed wrapper and the DataLoader for the three
second-stage datasets of the reference's scripts (``--dataset NTUMPII | NTUCOCO | NTUSeg --modal RGBD2S``,
datasets/ntu_mpii.py)."""
import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.distributed import DistributedSampler
from torch.utils.data.sampler import WeightedRandomSampler


class _SamplerAsDataset(Dataset):
    def __init__(self, sampler):
        self.sampler, self.drawn = sampler, None

    def __len__(self):
        return len(self.sampler)

    def __getitem__(self, i):
        if self.drawn is None:
            self.drawn = list(self.sampler)
        return self.drawn[i]


class DistributedSamplerWrapper(DistributedSampler):
    """Shards the indices DRAWN by another sampler across ranks (util.py:485-528): each epoch the wrapped
    sampler is drawn once per process, DistributedSampler picks this rank's positions of that draw."""

    def __init__(self, sampler, num_replicas=None, rank=None, shuffle=True):
        super().__init__(_SamplerAsDataset(sampler), num_replicas=num_replicas, rank=rank, shuffle=shuffle)
        self.sampler = sampler

    def __iter__(self):
        self.dataset.drawn = list(self.sampler)
        return iter([self.dataset.drawn[i] for i in super().__iter__()])


def source_balancing_weights(n_first, n_second):
    """Per-sample weights that give the two concatenated sources equal mass (util.py:558-581: MPII first)."""
    n = n_first + n_second
    w = np.zeros([n])
    w[:n_first] = n_second / n
    w[n_first:] = n_first / n
    return w


def build_own_contrast_loader(opt, rank=0, world=1, ngpus_per_node=1):
    """(dataset, loader, sampler) as ``datasets/util.py:530-585``; ``--batch_size`` is the GLOBAL batch."""
    key = (opt.dataset or '') + opt.modal
    from .ntu_mpii import NTUCOCOContrastDataset, NTUMPIIContrastDataset
    size = int(getattr(opt, 'image_size', 320))
    if key == 'NTUMPIIRGBD2S':
        ds = NTUMPIIContrastDataset(opt.data_folder, opt.train_file_list, opt.mpii_root, 'train', size=size,
                                    random_flip=bool(opt.random_flip), random_resized_crop=True)
    elif key == 'NTUCOCORGBD2S':
        ds = NTUCOCOContrastDataset(opt.data_folder, opt.train_file_list, opt.coco_root, 'train2014', size=size,
                                    random_flip=bool(opt.random_flip), random_resized_crop=True)
    elif key == 'NTUSegRGBD2S':
        from .ntu_mpii import NTUSegContrastDataset
        ds = NTUSegContrastDataset(opt.data_folder, opt.train_file_list, opt.seg_root, opt.seg_file_list, size=size,
                                   random_flip=bool(opt.random_flip), random_resized_crop=True,
                                   mask_seg_depth=bool(opt.mask_seg_depth), mask_seg_rgb=bool(opt.mask_seg_rgb))
    else:
        raise NotImplementedError('dataset %r: NTUMPII, NTUCOCO and NTUSeg (+ RGBD2S) have tuple producers in this build; '
                                  'use --synthetic otherwise' % key)
    if key == 'NTUSegRGBD2S':       # NTU frames first, parsing frames second (util.py:575-577)
        weights = source_balancing_weights(ds.split, len(ds) - ds.split)
    else:
        weights = source_balancing_weights(len(ds.db), len(ds.image_list))
    sampler = WeightedRandomSampler(weights, len(weights))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        sampler = DistributedSamplerWrapper(sampler, num_replicas=world, rank=rank)
    else:
        sampler.set_epoch = lambda epoch: None
    workers = int((opt.num_workers + ngpus_per_node - 1) / ngpus_per_node)
    loader = torch.utils.data.DataLoader(ds, batch_size=max(1, int(opt.batch_size / max(1, world))), shuffle=False,
                                         num_workers=workers, pin_memory=True, sampler=sampler, drop_last=True)
    return ds, loader, sampler
