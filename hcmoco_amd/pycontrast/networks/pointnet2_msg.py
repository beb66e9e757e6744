"""PointNet++ (multi-scale grouping) depth encoder of the HRNetPN arch
(/root/reference/pycontrast/networks/pointnet2_msg.py:10-95): 4 SA-MSG + 4 FP levels,
per-point features [B, 128, N]."""
import torch
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

    def plan(self, pointcloud, events=False):
        """Everything the forward pass derives from the coordinates alone -- FPS picks, centres and ball indices of the four
        SA levels, the three nearest neighbours + weights of the four FP levels -- as a GeometryPlan.  None of it depends on
        the features, so a caller may run it on a HIP stream of its own ahead of the feature path (``events=True`` records
        one event per SA level and one behind the FP neighbours; forward waits on them where it first needs each part):
        the FPS rounds are serial chains on 32 CUs that overlap with anything."""
        xyz, _ = self._break_up_pc(pointcloud)
        plan = GeometryPlan(xyz)
        for sa in self.SA_modules:
            plan.sa.append(sa.geometry(plan.l_xyz[-1]))
            plan.l_xyz.append(plan.sa[-1][0])
            plan.sa_ready.append(plan.mark(events))
        for k in range(len(self.FP_modules)):        # FP k: unknown = level k, known = level k + 1
            plan.fp.append(PointnetFPModule.neighbours(plan.l_xyz[k], plan.l_xyz[k + 1]))
        plan.fp_ready = plan.mark(events)
        return plan

    def forward(self, pointcloud, plan=None):
        if plan is not None:
            plan.wait(plan.sa_ready[0])        # the cloud itself may come from the plan's stream
        xyz, features = self._break_up_pc(pointcloud)
        if plan is not None:
            xyz = plan.l_xyz[0]
        l_xyz, l_features = [xyz], [features]
        for k, sa in enumerate(self.SA_modules):
            if plan is not None and k > 0:
                plan.wait(plan.sa_ready[k])
            nx, nf = sa(l_xyz[-1], l_features[-1], geometry=plan.sa[k] if plan is not None else None)
            l_xyz.append(nx)
            l_features.append(nf)
        if plan is not None:
            plan.wait(plan.fp_ready)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i],
                                                   neighbours=plan.fp[i] if plan is not None else None)
        return l_features[0]


class GeometryPlan:
    """Output of Pointnet2MSG.plan.  Tensors made on one stream and read on another are registered with the caching
    allocator through ``share`` (record_stream), as torch asks for."""

    def __init__(self, xyz):
        self.l_xyz, self.sa, self.sa_ready, self.fp, self.fp_ready = [xyz], [], [], [], None
        self.extra = {}

    @staticmethod
    def mark(events):
        if not events:
            return None
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @staticmethod
    def wait(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def tensors(self):
        for new_xyz, idx, offsets in self.sa:
            yield new_xyz
            for t in list(idx) + list(offsets):
                if t is not None:
                    yield t
        for idx, weight in self.fp:
            yield idx
            yield weight
        for v in self.extra.values():
            for t in (v if isinstance(v, (tuple, list)) else (v,)):
                if torch.is_tensor(t):
                    yield t
        yield self.l_xyz[0]

    def share(self, stream):
        for t in self.tensors():
            t.record_stream(stream)


def get_model(input_channels=0, class_num=1):
    return Pointnet2MSG(input_channels=input_channels, class_num=class_num)
