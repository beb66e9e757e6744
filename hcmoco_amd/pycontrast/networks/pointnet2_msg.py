"""PointNet++ (multi-scale grouping) depth encoder of the HRNetPN arch
(/root/reference/pycontrast/networks/pointnet2_msg.py:10-95): 4 SA-MSG + 4 FP levels,
per-point features [B, 128, N]."""
import torch.nn as nn

from .pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

NPOINTS = [4096, 1024, 256, 64]
RADIUS = [[0.025, 0.125], [0.125, 0.25], [0.25, 0.5], [0.5, 1.0]]
NSAMPLE = [[16, 32], [16, 32], [16, 32], [16, 32]]
MLPS = [[[16, 32], [32, 64]], [[64, 128], [64, 128]], [[128, 256], [128, 256]], [[256, 512], [256, 512]]]
FP_MLPS = [[128, 128], [256, 256], [512, 512], [512, 512]]


class Pointnet2MSG(nn.Module):
    def __init__(self, input_channels=6, class_num=1):
        super().__init__()
        self.class_num = class_num
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels
        skip = [input_channels]
        for k in range(len(NPOINTS)):
            mlps = [[channel_in] + list(m) for m in MLPS[k]]
            self.SA_modules.append(PointnetSAModuleMSG(npoint=NPOINTS[k], radii=RADIUS[k], nsamples=NSAMPLE[k],
                                                       mlps=mlps, use_xyz=True, bn=True))
            channel_in = sum(m[-1] for m in mlps)
            skip.append(channel_in)
        self.FP_modules = nn.ModuleList()
        for k in range(len(FP_MLPS)):
            pre = FP_MLPS[k + 1][-1] if k + 1 < len(FP_MLPS) else channel_in
            self.FP_modules.append(PointnetFPModule(mlp=[pre + skip[k]] + FP_MLPS[k]))

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud):
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        for sa in self.SA_modules:
            nx, nf = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(nx)
            l_features.append(nf)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        return l_features[0]


def get_model(input_channels=0, class_num=1):
    return Pointnet2MSG(input_channels=input_channels, class_num=class_num)
