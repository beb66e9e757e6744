"""On-device synthetic batch source with the positional tuple the trainer indexes
(reference producers: datasets/dataset.py:570-617).  Distributions follow
SURVEY.md 8d "config 2":

  rgb ~ N(0,1); depth ~ N(0,1) on a centred disc of radius 0.375*size, 0 elsewhere, x3 channels;
  depth_mask = that disc; skeleton ~ U(-1,1); original_joints2d ~ U(0,size) with 5% pushed out of
  range; joints_vis ~ Bernoulli(0.85); use_depth ~ Bernoulli(0.75) with at least one 1 (samples
  without depth get zero depth and an empty mask, dataset.py:574-575); index = a slice of a
  random permutation of the bank rows, disjoint across ranks.

A small pool of batches is generated once and stays resident in HBM, so a timed training step
contains no data generation and no host->device copy.
"""
import torch


class SyntheticContrastData(object):
    def __init__(self, n_data, batch_size, size=256, joints=16, steps=50, device='cpu', rank=0, world=1,
                 seed=0, pool=4, p_depth=0.75, ntu=False, p_rgb=1.0):
        self.n_data, self.batch_size, self.size, self.joints = n_data, batch_size, size, joints
        self.ntu = ntu               # append the NTU-only items 9-15 (needed by the HRNetPN arch)
        self.p_rgb = p_rgb           # P(use_rgb = 1) in NTU mode
        self.steps, self.device, self.rank, self.world = steps, torch.device(device), rank, world
        self.pool = [self._make(seed * 1000003 + i, p_depth) for i in range(pool)]

    def __len__(self):                      # `len(train_dataset)` sizes the bank (main_contrast.py:49)
        return self.n_data

    def _make(self, seed, p_depth):
        g = torch.Generator().manual_seed(seed)          # same stream on every rank, sliced per rank below
        B, W, H, J = self.batch_size, self.world, self.size, self.joints
        lo, hi = self.rank * B, (self.rank + 1) * B
        index = torch.randperm(self.n_data, generator=g)[:B * W][lo:hi]
        gl = torch.Generator().manual_seed(seed * 7919 + self.rank + 1)
        use_depth = (torch.rand(B, generator=gl) < p_depth).long()
        use_depth[0] = 1
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing='ij')
        disc = (((yy - H / 2) ** 2 + (xx - H / 2) ** 2) < (0.375 * H) ** 2).float()
        mask = disc.unsqueeze(0) * use_depth.view(B, 1, 1).float()
        rgb = torch.randn(B, 3, H, H, generator=gl)
        depth = (torch.randn(B, 1, H, H, generator=gl) * mask.unsqueeze(1)).expand(B, 3, H, H)
        skeleton = torch.rand(B, J, 2, generator=gl) * 2 - 1
        j2d = torch.rand(B, J, 2, generator=gl) * H
        out_of_range = torch.rand(B, J, generator=gl) < 0.05
        j2d = torch.where(out_of_range.unsqueeze(-1), j2d * 1.3 - 0.15 * H, j2d)
        vis = (torch.rand(B, J, generator=gl) < 0.85).int()
        batch = [torch.cat([rgb, depth], 1).contiguous(), index, skeleton, torch.zeros(B, 25, 3), j2d, vis,
                 use_depth, mask, torch.ones(B)]
        if self.ntu:
            # identity pixel grid of a 1080x1920 frame cropped to a centred person box and
            # nearest-resized to size x size; per-sample mean depth in metres
            ys = torch.linspace(240, 840, H).round().int()
            xs = torch.linspace(660, 1260, H).round().int()
            gy, gx = torch.meshgrid(ys, xs, indexing='ij')
            grid_xy = torch.stack([gy, gx], -1).unsqueeze(0).expand(B, H, H, 2).contiguous()
            mean = torch.rand(B, generator=gl) * 2 + 2
            # item 11 = true_rgb -> use_rgb (datasets/dataset.py:1083-1103): NTU frames can lack the RGB modality
            use_rgb = (torch.rand(B, generator=gl) < self.p_rgb).long()
            use_rgb[0] = 1                               # sample 0 keeps both modalities (use_depth[0] = 1 too)
            batch += [torch.zeros(B, H, H, dtype=torch.long), torch.zeros(B, dtype=torch.long), use_rgb,
                      grid_xy, torch.full((B,), 1080), torch.full((B,), 1920), mean]
        return [t.to(self.device) for t in batch]

    def __iter__(self):
        for i in range(self.steps):
            yield self.pool[i % len(self.pool)]


class _Loader(object):
    def __init__(self, data):
        self.data = data

    def __len__(self):
        return self.data.steps

    def __iter__(self):
        return iter(self.data)


class _Sampler(object):
    def set_epoch(self, epoch):
        pass


def build_synthetic_contrast_loader(opt, device, rank=0, world=1):
    """(dataset, loader, sampler) like ``build_own_contrast_loader`` (datasets/util.py:530-585);
    ``--batch_size`` is the GLOBAL batch there (:539), so each rank takes batch_size // world."""
    from ..networks.sgcn import num_joints
    per_rank = max(1, opt.batch_size // max(1, world))
    data = SyntheticContrastData(opt.synthetic_n_data, per_rank, opt.synthetic_size, num_joints(opt.skeleton_meta_name),
                                 opt.synthetic_steps, device, rank, world, seed=opt.seed or 0,
                                 ntu=(opt.arch == 'HRNetPN' or bool(getattr(opt, 'synthetic_ntu', 0))),
                                 p_rgb=float(getattr(opt, 'synthetic_p_rgb', 1.0)))
    return data, _Loader(data), _Sampler()
