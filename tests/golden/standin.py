"""Tiny stand-in encoder with the forward contract of the RGBD2S HRNet model
(networks/build_backbone.py:256-303 of the reference): two 1x1 "encoders" on the 4x-pooled RGB / depth
images, a linear keypoint encoder, three linear + L2 heads.  Written for this repo (it is NOT reference
code); ``gen_golden.py`` drives the reference's own training loops with it to record the 2-step traces
of SURVEY 8c, and the tests rebuild it from the recorded initial weights."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class StandInEncoder(nn.Module):
    def __init__(self, C=128, D=128):
        super().__init__()
        self.proj1 = nn.Conv2d(3, C, 1)
        self.proj2 = nn.Conv2d(3, C, 1)
        self.graph = nn.Linear(2, C)
        self.head1 = nn.Linear(C, D)
        self.head2 = nn.Linear(C, D)
        self.head3 = nn.Linear(C, D)

    def forward(self, x, s, mode=0, return_fm=False):
        m1 = self.proj1(F.avg_pool2d(x[:, :3], 4))
        m2 = self.proj2(F.avg_pool2d(x[:, 3:], 4))
        feat3 = self.graph(s)
        f = torch.cat([F.normalize(self.head1(m1.mean((2, 3)))), F.normalize(self.head2(m2.mean((2, 3)))),
                       F.normalize(self.head3(feat3.mean(1)))], dim=1)
        if not return_fm:
            return f
        return [m1], [m2], feat3, f, {'merge1': m1, 'merge2': m2, 'linear_merge1': m1, 'linear_merge2': m2}


def make_batches(steps, B, n, H, J, seed):
    """Positional batch tuples (NTU-style with use_rgb at position 11).
    Step t+1 re-uses two bank rows of step t, so it must see the rows step t wrote."""
    g = torch.Generator().manual_seed(seed)
    out, prev = [], None
    for t in range(steps):
        index = torch.randperm(n, generator=g)[:B]
        if prev is not None:
            index[0], index[3] = prev[1], prev[4]
        index[B - 1] = index[2]                         # duplicate inside the batch: last one wins
        prev = index.clone()
        use_depth = torch.tensor([1, 1, 0, 1, 1, 0][:B] + [1] * max(0, B - 6))
        use_rgb = torch.tensor([1, 0, 1, 1, 1, 1][:B] + [1] * max(0, B - 6))
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing='ij')
        disc = (((yy - H / 2) ** 2 + (xx - H / 2) ** 2) < (0.4 * H) ** 2).float()
        mask = disc.unsqueeze(0) * use_depth.view(B, 1, 1).float()
        rgb = torch.randn(B, 3, H, H, generator=g)
        depth = (torch.randn(B, 1, H, H, generator=g) * mask.unsqueeze(1)).expand(B, 3, H, H)
        skeleton = torch.rand(B, J, 2, generator=g) * 2 - 1
        j2d = torch.rand(B, J, 2, generator=g) * H
        j2d[0, 0] = torch.tensor([-2.0, H + 3.0])
        vis = (torch.rand(B, J, generator=g) < 0.85).int()
        vis[:, 0] = 1
        out.append([torch.cat([rgb, depth], 1).contiguous(), index, skeleton, torch.zeros(B, 25, 3), j2d, vis,
                    use_depth, mask, torch.ones(B), torch.zeros(B, H, H, dtype=torch.long),
                    torch.zeros(B, dtype=torch.long), use_rgb])
    return out


class StandInMoCoEncoder(nn.Module):
    """Tiny stand-in with the forward contract of the reference's CMC single-head model
    (networks/build_backbone.py:72-120: ``model(x, mode)`` -> [B, 2 D], an L and an ab half): 1x1 conv + BatchNorm +
    ReLU + pool + linear + L2 per half.  The BatchNorm is what makes the MoCo loop's shuffled key batch matter.
    Written for this repo (NOT reference code); drives the reference's ``_train_moco`` in gen_golden.py."""

    def __init__(self, C=16, D=32):
        super().__init__()
        self.conv_l, self.conv_ab = nn.Conv2d(1, C, 1), nn.Conv2d(2, C, 1)
        self.bn_l, self.bn_ab = nn.BatchNorm2d(C), nn.BatchNorm2d(C)
        self.head_l, self.head_ab = nn.Linear(C, D), nn.Linear(C, D)

    def forward(self, x, x_jig=None, mode=0):
        fl = F.relu(self.bn_l(self.conv_l(x[:, :1]))).mean((2, 3))
        fab = F.relu(self.bn_ab(self.conv_ab(x[:, 1:3]))).mean((2, 3))
        if mode in (0, 1):
            return torch.cat([F.normalize(self.head_l(fl)), F.normalize(self.head_ab(fab))], dim=1)
        return torch.cat([fl, fab], dim=1)


def make_moco_batches(steps, B, H, seed):
    """(two 3-channel crops stacked on the channel axis [B, 6, H, H], index) per step, like the CMC loader."""
    g = torch.Generator().manual_seed(seed)
    return [[torch.randn(B, 6, H, H, generator=g), torch.arange(B) + t * B] for t in range(steps)]
