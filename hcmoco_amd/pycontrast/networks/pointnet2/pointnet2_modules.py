"""Set-abstraction (multi-scale grouping) and feature-propagation modules
(/root/reference/pycontrast/networks/pointnet2/pointnet2_modules.py:10-156)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class PointnetSAModuleMSG(nn.Module):
    """FPS -> gather centres -> per scale: ball query, group, shared MLP, max-pool over the ball."""

    def __init__(self, *, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))

    @staticmethod
    def ball_offsets(xyz_t, new_xyz, idx):
        """xyz[idx] - centre -> [B, 3, np, ns]: QueryAndGroup's grouped_xyz (pointnet2_utils.py:251-253 of the reference),
        the same gather-then-subtract, so the same fp32 values.  No autograd: the clouds carry no gradient."""
        with torch.no_grad():
            return pointnet2_utils.grouping_operation(xyz_t, idx).sub_(new_xyz.transpose(1, 2).unsqueeze(-1))

    def _offsets_wanted(self, xyz):
        """The scales whose first layer runs on the implicit grouped tensor need the relative offsets of their balls."""
        return [self.npoint is not None and g.use_xyz and len(mlp) > 1 and g.nsample in (4, 8, 16, 32, 64)
                and self.npoint % 4 == 0 and pt_utils.first_layer_fusable(mlp, xyz)
                for g, mlp in zip(self.groupers, self.mlps)]

    def geometry(self, xyz, new_xyz=None):
        """The feature-independent half of forward: FPS picks -> centres, ball indices of every scale and (for the
        scales that take the fused first layer) the members' offsets from their centre."""
        xyz_t = None
        if new_xyz is None and self.npoint is not None:
            picks = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            xyz_t = xyz.transpose(1, 2).contiguous()
            new_xyz = pointnet2_utils.gather_operation(xyz_t, picks).transpose(1, 2).contiguous()
        idx = [pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz) if self.npoint is not None else None
               for g in self.groupers]
        offsets = [None] * len(idx)
        for i, want in enumerate(self._offsets_wanted(xyz)):
            if want and new_xyz.shape[1] % 4 == 0:
                xyz_t = xyz.transpose(1, 2).contiguous() if xyz_t is None else xyz_t
                offsets[i] = self.ball_offsets(xyz_t, new_xyz, idx[i])
        return new_xyz, idx, offsets

    def forward(self, xyz, features=None, new_xyz=None, geometry=None):
        """``geometry``: the (new_xyz, idx per scale) pair of ``self.geometry(xyz)`` when it was computed ahead."""
        if geometry is not None:
            new_xyz, ball_idx, ball_off = geometry
        else:
            ball_idx = [None] * len(self.groupers)
            ball_off = [None] * len(self.groupers)
            if new_xyz is None and self.npoint is not None:
                picks = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
                new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), picks).transpose(1, 2).contiguous()
        outs = []
        # the centre count is the CALLER's when new_xyz / geometry is handed in: gate on it, not on self.npoint
        centres = new_xyz.shape[1] if new_xyz is not None else 0
        wanted = self._offsets_wanted(xyz)
        for grouper, mlp, bidx, off, want in zip(self.groupers, self.mlps, ball_idx, ball_off, wanted):
            layers = list(mlp)
            fuse_first = want and centres % 4 == 0
            if fuse_first and bidx is None:
                bidx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
            if fuse_first and off is None:
                off = self.ball_offsets(xyz.transpose(1, 2).contiguous(), new_xyz, bidx)
            if fuse_first:
                # r05: the first layer on the IMPLICIT grouped tensor -- [B, 3 + C, npoint, nsample] is never built
                x = layers[0].forward_grouped(off, features, bidx)
                mlp = layers[1:]
            else:
                x = grouper(xyz, new_xyz, features, idx=bidx)               # (B, C, npoint, nsample)
            if self.pool_method == 'max_pool' and pt_utils.ballmax_fusable(mlp, x):
                # r05: the last conv -> BatchNorm -> ReLU and the max over the ball as one node; relu(bn(z)) is never
                # written or re-read (4 passes over the largest tensors of the network instead of 12)
                for layer in list(mlp)[:-1]:
                    x = layer(x)
                outs.append(list(mlp)[-1].forward_ballmax(x))
                continue
            y = x
            for layer in mlp:
                y = layer(y)
            if self.pool_method == 'max_pool':
                if y.is_cuda and y.dtype == torch.float32:   # hcm_rowmax_*: same values, same (first-index) tie rule
                    from .... import pointnet2_hip
                    outs.append(pointnet2_hip.ball_max(y))
                    continue
                y = F.max_pool2d(y, kernel_size=[1, y.size(3)])
            elif self.pool_method == 'avg_pool':
                y = F.avg_pool2d(y, kernel_size=[1, y.size(3)])
            else:
                raise NotImplementedError
            outs.append(y.squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """inverse-distance interpolation from the 3 nearest known points, skip concat, shared MLP."""

    def __init__(self, *, mlp, bn=True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def neighbours(unknown, known):
        """three nearest known points of every unknown point and their inverse-distance weights (feature-independent)."""
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        dist_recip = 1.0 / (dist + 1e-8)
        return idx, dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)

    def forward(self, unknown, known, unknow_feats, known_feats, neighbours=None):
        if known is not None:
            idx, weight = neighbours if neighbours is not None else self.neighbours(unknown, known)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        feats = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(feats.unsqueeze(-1)).squeeze(-1)
