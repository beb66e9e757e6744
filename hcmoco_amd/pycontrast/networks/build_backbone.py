"""Encoder plugin surface: ``build_model(opt) -> (model, model_ema)``.

Reference: /root/reference/pycontrast/networks/build_backbone.py:186-303 (RGBD2S HRNet model),
:516-566 (registry + factory).  Registry key = ``opt.modal + opt.arch + ('Mul' if jigsaw else
'Sin')``; constructors take the same positional arguments; ``state_dict`` top-level names are
``encoder1, encoder2, encoder3, head1-3, encoder1_linear, encoder2_linear``.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .hrnet import HighResolutionNet, upsample_bilinear
from .sgcn import create_sgcn
from .util import Normalize


class CMC3HRNetSGCNSingleHead(nn.Module):
    """RGB HRNet + depth HRNet + SemGCN keypoint encoder, one linear+L2 head each."""

    def __init__(self, name='HRNet', head='linear', feat_dim=128, in_channel_list=(3, 3, 3),
                 linear_feat_map=False, width=18, pool_method='mean', opt=None):
        super().__init__()
        assert name == 'HRNet'
        assert pool_method in ('mean', 'max')
        if width not in (18, 32, 48):
            raise NotImplementedError(width)
        if head != 'linear':
            raise NotImplementedError('head not supported: {}'.format(head))
        self.opt = opt
        self.in_channel_list = list(in_channel_list)
        self.linear_feat_map = linear_feat_map
        self.width = width
        self.pool_method = pool_method
        dim_in = sum(width * 2 ** i for i in range(4))
        sgcn_dim = 128
        self.encoder1 = HighResolutionNet(width)
        self.encoder2 = HighResolutionNet(width)
        self.encoder3 = create_sgcn(opt.skeleton_meta_name, sgcn_dim, 4)
        self.head1 = nn.Sequential(nn.Linear(dim_in, feat_dim), Normalize(2))
        self.head2 = nn.Sequential(nn.Linear(dim_in, feat_dim), Normalize(2))
        self.head3 = nn.Sequential(nn.Linear(sgcn_dim, feat_dim), Normalize(2))
        if self.linear_feat_map:
            self.encoder1_linear = nn.Conv2d(dim_in, sgcn_dim, kernel_size=1, stride=1, bias=True)
            self.encoder2_linear = nn.Conv2d(dim_in, sgcn_dim, kernel_size=1, stride=1, bias=True)
        # Set by a trainer whose loss engine projects only the sampled pixels (engine.fmap_sampled):
        # ``return_fm`` then hands back the raw branch maps and skips merge_all_res + the full 1x1
        # projections (aux entries are None).  Off by default = the reference data flow.
        self.defer_projection = False
        # Set by a trainer whose loss engine also computes the pooling + heads inside its fused loss section
        # (engine.section, csrc/section.hip): ``return_fm`` then returns ``f = None`` next to the raw branch maps.
        self.defer_heads = False
        # The three encoders are independent until the heads.  ``two_streams`` (an attribute; tests and the quiet first step
        # set it to 0) is a bit mask:
        #   1            SemGCN on a side HIP stream, issued first: its single-workgroup kernels
        #                (3.5 ms per step, one CU) run underneath the HRNets instead of in line;
        #   2            encoder2 on a second side stream (small HRNet kernels overlap; when it runs as a
        #                compiled program its forward is issued by that stream's C++ helper thread);
        #                (Issuing encoder2 from a PYTHON helper thread was measured slower -- the GIL -- and is gone.)
        # Default 2 since r03: the SemGCN layers are batch-parallel kernels now (0.6 ms per step, r02) and run on the
        # caller's stream right behind encoder1, underneath encoder2's tail; a stream of their own is one more active
        # stream for four hardware queues to share (697-699 -> 699-702 samples/s, alternating runs on one box).
        # Backward follows automatically: autograd replays every node on its forward stream.
        self.two_streams = 2
        self._side_streams = {}
        # torch.bfloat16: the two HRNets run under bf16 autocast (module path: stock MIOpen bf16 convolutions, batch
        # norm with fp32 statistics and parameters); their maps come back as fp32.  Set by the trainer from
        # --encoder_dtype / --amp.
        self.encoder_dtype = torch.float32

    def _run_hrnet(self, enc, x):
        if self.encoder_dtype == torch.float32 or not x.is_cuda:
            return enc(x)
        with torch.autocast('cuda', dtype=self.encoder_dtype):
            return [m.float() for m in enc(x.to(self.encoder_dtype))]

    def _side(self, idx, device):
        st = self._side_streams.get((idx, device))
        if st is None:
            st = self._side_streams[(idx, device)] = torch.cuda.Stream(device=device)
        return st

    def _encode(self, x1, x2, s):
        """(feat1 maps, feat2 maps, feat3) with the stream placement described in __init__."""
        mode = self.two_streams if x1.is_cuda else 0
        if not mode:
            return self._run_hrnet(self.encoder1, x1), self._run_hrnet(self.encoder2, x2), self.encoder3(s)
        mixed = self.encoder_dtype != torch.float32
        main = torch.cuda.current_stream(x1.device)
        joined = []

        def on_side(idx, fn):
            side = self._side(idx, x1.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out = fn()
            joined.append((side, out))
            return out

        feat3 = on_side(0, lambda: self.encoder3(s)) if mode & 1 else None
        if mode & 2:
            # encoder2 on its own stream; when it runs as a compiled program (networks/hrnet.py) its
            # forward is issued by that stream's C++ helper thread while this thread issues encoder1
            handle = []

            def start():
                h = self.encoder2.forward_async(x2) if (hasattr(self.encoder2, 'forward_async') and not mixed) else None
                handle.append(h)
                return [] if h is not None else self._run_hrnet(self.encoder2, x2)

            feat2 = on_side(1, start)
            feat1 = self._run_hrnet(self.encoder1, x1)
            if handle[0] is not None:
                feat2.extend(self.encoder2.forward_wait(handle[0]))
        else:
            feat1, feat2 = self._run_hrnet(self.encoder1, x1), self._run_hrnet(self.encoder2, x2)
        if feat3 is None:
            feat3 = self.encoder3(s)
        for side, out in joined:             # consumed on the main stream from here on
            main.wait_stream(side)
            for t in (out if isinstance(out, (list, tuple)) else (out,)):
                t.record_stream(main)
        return feat1, feat2, feat3

    @staticmethod
    def merge_all_res(maps):
        """Upsample the three coarser maps to the finest grid and concatenate (:247-254)."""
        size = maps[0].shape[-2:]
        ups = [maps[0]] + [upsample_bilinear(m, size) for m in maps[1:]]
        return torch.cat(ups, 1)

    def _pool(self, maps):
        red = torch.amax if self.pool_method == 'max' else torch.mean
        return torch.cat([red(m, dim=(2, 3)) for m in maps], 1)

    def forward(self, x, s, mode=0, return_fm=False):
        """mode 0/1: projected + L2-normalised features; 2: raw pooled features (:256-303)."""
        x1, x2 = torch.split(x, self.in_channel_list, dim=1)
        _feat1, _feat2, _feat3 = self._encode(x1, x2, s)
        if (self.defer_heads and self.defer_projection and return_fm and self.linear_feat_map and mode in (0, 1)
                and self.pool_method == 'mean'):
            return _feat1, _feat2, _feat3, None, {'merge1': None, 'merge2': None, 'linear_merge1': None,
                                                  'linear_merge2': None}
        avg1, avg2, avg3 = self._pool(_feat1), self._pool(_feat2), _feat3.mean(1)
        if mode in (0, 1):
            feat1, feat2, feat3 = self.head1(avg1), self.head2(avg2), self.head3(avg3)
        else:
            feat1, feat2, feat3 = avg1, avg2, avg3
        f = torch.cat((feat1, feat2, feat3), dim=1)
        if not return_fm:
            return f
        if self.linear_feat_map:
            if self.defer_projection:
                return _feat1, _feat2, _feat3, f, {'merge1': None, 'merge2': None,
                                                   'linear_merge1': None, 'linear_merge2': None}
            merge1, merge2 = self.merge_all_res(_feat1), self.merge_all_res(_feat2)
            return _feat1, _feat2, _feat3, f, {
                'merge1': merge1, 'merge2': merge2,
                'linear_merge1': self.encoder1_linear(merge1),
                'linear_merge2': self.encoder2_linear(merge2),
            }
        return _feat1, _feat2, _feat3, avg1, avg2, avg3, f


class CMC3HRNetSGCNPN2SingleHead(nn.Module):
    """``HRNetPN`` arch: RGB HRNet + PointNet++ (MSG) on the back-projected depth cloud + SemGCN
    (build_backbone.py:305-514).  The nine point ops run in the HIP kernels of this repo
    (networks/pointnet2/pointnet2_utils.py -> hcmoco_amd.pointnet2_hip)."""

    NUM_POINTS = 4096

    def __init__(self, name='HRNetPN', head='linear', feat_dim=128, in_channel_list=(3, 3, 3),
                 linear_feat_map=False, width=18, pool_method='mean', opt=None):
        super().__init__()
        assert name == 'HRNetPN'
        assert pool_method in ('mean', 'max')
        if width not in (18, 32, 48):
            raise NotImplementedError(width)
        if head != 'linear':
            raise NotImplementedError('head not supported: {}'.format(head))
        from .pointnet2_msg import Pointnet2MSG
        from .pointnet2 import pytorch_utils as pt_utils
        self.opt = opt
        self.in_channel_list = list(in_channel_list)
        self.linear_feat_map = linear_feat_map
        self.width = width
        self.pool_method = pool_method
        dim_in = sum(width * 2 ** i for i in range(4))
        self.encoder1 = HighResolutionNet(width)
        self.encoder2 = Pointnet2MSG(input_channels=0)
        self.pn_dim = 128
        sgcn_dim = 128
        self.encoder3 = create_sgcn(opt.skeleton_meta_name, sgcn_dim, 4)
        self.head1 = nn.Sequential(nn.Linear(dim_in, feat_dim), Normalize(2))
        self.two_streams = 7          # bit 0: SemGCN, bit 1: the cloud branch, bit 2: the cloud's geometry on streams of their own
        self.trace_streams = False    # tools/probes/hrnetpn_streams.py: record when each branch starts / ends on the GPU
        self._side_streams = {}
        self.head2 = nn.Sequential(nn.Linear(self.pn_dim, feat_dim), Normalize(2))
        self.head3 = nn.Sequential(nn.Linear(sgcn_dim, feat_dim), Normalize(2))
        if self.linear_feat_map:
            self.encoder1_linear = nn.Conv2d(dim_in, sgcn_dim, kernel_size=1, stride=1, bias=True)
            self.encoder2_linear = pt_utils.Conv1d(self.pn_dim, sgcn_dim, bn=True)
        # set by a trainer whose loss engine runs pooling, heads, merge_all_res + encoder1_linear at the sampled pixels
        # and the losses as one node (engine.section -> hip_ops.stage2_section_pn): ``return_fm`` then returns f = None,
        # the raw HRNet branch maps, the cloud features and the depth map in aux['linear_merge2']
        self.defer_projection = False
        self.defer_heads = False

    merge_all_res = staticmethod(CMC3HRNetSGCNSingleHead.merge_all_res)
    _pool = CMC3HRNetSGCNSingleHead._pool
    _side = CMC3HRNetSGCNSingleHead._side

    def depth2pts(self, depth, depth_mask, grid_xy, ori_h, ori_w, mean):
        """Back-project every pixel (X=(gx-H0/2) z k, Y=(W0/2-gy) z k, Z=z, k=0.0035; z = depth+mean,
        re-centred) and draw NUM_POINTS valid pixels per image with replacement (:379-445).
        Images with an empty mask keep all-zero clouds.  Sync-free (no boolean-mask indexing)."""
        bs, size = depth.shape[0], depth.shape[-1]
        mean = mean.reshape(bs, 1, 1).to(depth.dtype)
        gx = grid_xy[..., 0].reshape(bs, size, size).float()
        gy = grid_xy[..., 1].reshape(bs, size, size).float()
        z = depth[:, 0] + mean
        xyz = torch.stack([(gx - ori_h / 2) * z * 0.0035, (ori_w / 2 - gy) * z * 0.0035, z - mean], 1)
        xyz = xyz.reshape(bs, 3, size * size).float()
        valid = F.interpolate(depth_mask.unsqueeze(1).float(), size=(size, size), mode='nearest').reshape(bs, -1)
        keep = valid.sum(-1) > 0
        ind = torch.multinomial(valid + (~keep).unsqueeze(1).to(valid.dtype), self.NUM_POINTS, replacement=True)
        keepf = keep.to(xyz.dtype).view(bs, 1, 1)
        sampled = torch.gather(xyz, 2, ind.unsqueeze(1).expand(bs, 3, self.NUM_POINTS)) * keepf
        return sampled, xyz * keepf, ind

    @staticmethod
    def pts2depth(sampled_pts, pts, feat, h, w):
        """Spread per-point features back to all H*W pixels by 3-NN inverse-distance weights (:447-455)."""
        from .pointnet2 import pointnet2_utils
        dist, idx = pointnet2_utils.three_nn(pts.transpose(1, 2).contiguous(), sampled_pts.transpose(1, 2).contiguous())
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        out = pointnet2_utils.three_interpolate(feat.contiguous(), idx, weight)
        return out.reshape(feat.shape[0], feat.shape[1], h, w)

    _kept_pixels = {}

    @classmethod
    def kept_pixels(cls, h, w, oh, ow, device):
        """Linear indices (into the h x w input) of the pixels ``F.interpolate(.., size=(oh, ow))`` (nearest) reads, obtained
        from F.interpolate itself on a grid of pixel numbers (exact in fp32 below 2^24), cached per shape."""
        key = (h, w, oh, ow, str(device))
        if key not in cls._kept_pixels:
            assert h * w < (1 << 24)
            grid = torch.arange(h * w, dtype=torch.float32, device=device).view(1, 1, h, w)
            cls._kept_pixels[key] = F.interpolate(grid, size=(oh, ow)).reshape(-1).long()
            if grid.is_cuda:
                # shared by every model (and model_ema) of the process and by every stream, never freed: finish it HERE,
                # once, so no later reader on another stream can get ahead of the kernels that fill it
                torch.cuda.current_stream(grid.device).synchronize()
        return cls._kept_pixels[key]

    @classmethod
    def pts2depth_resized(cls, sampled_pts, pts, feat, h, w, oh, ow):
        """``F.interpolate(pts2depth(sampled_pts, pts, feat, h, w), size=(oh, ow))`` (:299-300 of the reference) without the
        pixels the nearest resize throws away: at 256^2 -> 64^2 it keeps one pixel in sixteen, so three_nn, three_interpolate
        and their backward run on 4096 pixels per image instead of 65 536 and the [B, 128, 256, 256] map (1 GB at B = 32) is
        never written.  Every kept pixel is computed exactly as in the full map (three_nn / three_interpolate are
        per-pixel), and the discarded pixels carry exact-zero gradients in the full flow (tests/test_pointnet2_gpu.py compares
        with the two-step route)."""
        keep = cls.kept_pixels(h, w, oh, ow, pts.device)
        out = cls.pts2depth(sampled_pts, pts.index_select(2, keep), feat, oh, ow)
        return out

    @staticmethod
    def _stem_hw(n):
        """Side of the HRNet's first-branch map for an n-pixel side: two 3x3 stride-2 pad-1 convolutions (hrnet.py stem)."""
        n = (n - 1) // 2 + 1
        return (n - 1) // 2 + 1

    def forward(self, x, s, depth_mask, grid_xy, original_h, original_w, mean, mode=0, return_fm=False):
        x1, x2 = torch.split(x, self.in_channel_list, dim=1)
        h, w = x1.shape[-2:]
        want_map = return_fm and self.linear_feat_map
        oh, ow = self._stem_hw(h), self._stem_hw(w)
        linear_merge2 = None

        def cloud_branch():
            sample, full, _ = self.depth2pts(x2, depth_mask, grid_xy, original_h, original_w, mean)
            return sample, full, self.encoder2(sample.transpose(1, 2))           # [B, 128, 4096]

        def depth_map(sample, full, feat2, neighbours=None):
            # networks/build_backbone.py:299-300 of the reference, at the pixels the resize keeps (pts2depth_resized)
            lin = self.encoder2_linear(feat2)
            if neighbours is None:
                return self.pts2depth_resized(sample, full, lin, h, w, oh, ow)
            from .pointnet2 import pointnet2_utils
            return pointnet2_utils.three_interpolate(lin.contiguous(), *neighbours).reshape(lin.shape[0], lin.shape[1], oh, ow)

        if x.is_cuda and self.two_streams:
            # the point-cloud branch (back-projection, PointNet++) and the SemGCN on side HIP streams,
            # the HRNet on the caller's: same placement rule as CMC3HRNetSGCNSingleHead._encode
            main = torch.cuda.current_stream(x.device)
            side_pn = self._side(1, x.device)
            side_g = self._side(0, x.device) if self.two_streams & 1 else None
            side_geo = self._side(2, x.device) if self.two_streams & 4 else None
            side_pn.wait_stream(main)
            trace = self.trace_streams
            if trace:
                ev = {k: torch.cuda.Event(enable_timing=True) for k in ('t0', 'h0', 'h1', 'p0', 'p1')}
                ev['t0'].record(main)
            # r05: the HRNet's forward goes to the C++ helper thread of the caller's stream (encoder_forward_async) FIRST -- the
            # geometry's ~40 Python-issued launches and the cloud branch's ~700 are issued while that thread issues the
            # program's ~360 -- and is collected BEHIND the cloud branch: encoder_forward_wait makes the program's autograd
            # node on this thread at that point, i.e. AFTER every node of the cloud branch, so the engine -- which runs the
            # ready node that was created last -- starts the HRNet's reverse loop (one push to its helper thread) before it
            # walks the cloud branch's ~700 nodes instead of after them (r04: the HRNet queue idle for the first ~15 ms of
            # backward).
            hr_pending = None
            early = side_geo is not None and hasattr(self.encoder1, 'forward_async')
            inputs_ready = torch.cuda.Event()          # the side streams wait for the inputs, not for the HRNet launches
            inputs_ready.record(main)                  # that the helper thread is about to queue behind them
            if early:
                if trace:
                    ev['h0'].record(main)
                hr_pending = self.encoder1.forward_async(x1, node_at_wait=True)
            if side_g is not None:
                side_g.wait_event(inputs_ready)
                with torch.cuda.stream(side_g):
                    _feat3 = self.encoder3(s)
            # r04: the GEOMETRY of the cloud branch (back-projection, the four FPS levels, eight ball queries, the three_nn of
            # the four FP levels and of pts2depth) depends on the depth input alone.  It goes first, on a stream of its own
            # (bit 2 of ``two_streams``): 40 launches of serial, low-occupancy kernels (an FPS level is 32 workgroups
            # walking a chain of rounds) that run underneath the HRNet instead of in front of the cloud branch's MLPs.
            plan = None
            if side_geo is not None:
                side_geo.wait_event(inputs_ready)
                with torch.cuda.stream(side_geo):
                    from .pointnet2.pointnet2_modules import PointnetFPModule
                    sample_pn, full_pn, _ = self.depth2pts(x2, depth_mask, grid_xy, original_h, original_w, mean)
                    cloud_in = sample_pn.transpose(1, 2)
                    plan = self.encoder2.plan(cloud_in, events=True)
                    plan.extra['cloud'] = (sample_pn, full_pn, cloud_in)
                    if want_map:
                        keep = self.kept_pixels(h, w, oh, ow, x.device)
                        plan.extra['map'] = PointnetFPModule.neighbours(
                            full_pn.index_select(2, keep).transpose(1, 2).contiguous(), sample_pn.transpose(1, 2).contiguous())
                        plan.extra['map_ready'] = plan.mark(True)
                    plan.share(side_pn)
                    plan.share(main)
            # Issue order.  The HRNet is one compiled program whose launches a C++ loop issues in ~2 ms; the cloud branch is
            # ~700 launches issued from Python.  r04: with the geometry on its own stream the HRNet goes before the cloud
            # branch (61.4 vs 62.1 ms per synchronised step).  r05: and before the geometry too, from its helper thread -- the
            # profiled step had both encoder queues idle for 5.7 ms while Python issued the geometry's ~150 small kernels:
            # same-box A/B 660.8 / 660.2 vs 649.6 samples/s (a cold 625 aside).
            first = plan is not None
            if first and hr_pending is None:
                if trace:
                    ev['h0'].record(main)
                hr_pending = self.encoder1.forward_async(x1, node_at_wait=True) if hasattr(self.encoder1, 'forward_async') else None
                if hr_pending is None:
                    _feat1 = self.encoder1(x1)
                    if trace:
                        ev['h1'].record(main)
            with torch.cuda.stream(side_pn):
                if trace:
                    ev['p0'].record(side_pn)
                if plan is None:
                    sample_pn, full_pn, _feat2 = cloud_branch()
                    if want_map:
                        linear_merge2 = depth_map(sample_pn, full_pn, _feat2)
                else:
                    _feat2 = self.encoder2(cloud_in, plan=plan)
                    if want_map:
                        plan.wait(plan.extra['map_ready'])
                        linear_merge2 = depth_map(sample_pn, full_pn, _feat2, plan.extra['map'])
                if trace:
                    ev['p1'].record(side_pn)
            if hr_pending is not None:
                _feat1 = self.encoder1.forward_wait(hr_pending)
                if trace:
                    ev['h1'].record(main)
            if not first:
                if trace:
                    ev['h0'].record(main)
                _feat1 = self.encoder1(x1)
                if trace:
                    ev['h1'].record(main)
            if trace:
                self._stream_trace = ev
            if side_g is None:
                _feat3 = self.encoder3(s)          # behind the HRNet on the caller's stream (see CMC3HRNetSGCNSingleHead)
            main.wait_stream(side_pn)
            if side_g is not None:
                main.wait_stream(side_g)
            for t in (sample_pn, full_pn, _feat2, _feat3, linear_merge2):
                if t is not None:
                    t.record_stream(main)
        else:
            _feat1 = self.encoder1(x1)
            sample_pn, full_pn, _feat2 = cloud_branch()
            _feat3 = self.encoder3(s)
            if want_map:
                linear_merge2 = depth_map(sample_pn, full_pn, _feat2)
        if (self.defer_heads and self.defer_projection and want_map and mode in (0, 1) and self.pool_method == 'mean'):
            return _feat1, _feat2, _feat3, None, {'merge1': None, 'merge2': _feat2, 'linear_merge1': None,
                                                  'linear_merge2': linear_merge2}
        avg1, avg2, avg3 = self._pool(_feat1), _feat2.mean(-1), _feat3.mean(1)
        if mode in (0, 1):
            feat1, feat2, feat3 = self.head1(avg1), self.head2(avg2), self.head3(avg3)
        else:
            feat1, feat2, feat3 = avg1, avg2, avg3
        f = torch.cat((feat1, feat2, feat3), dim=1)
        if not return_fm:
            return f
        if self.linear_feat_map:
            merge1 = self.merge_all_res(_feat1)
            linear_merge1 = self.encoder1_linear(merge1)
            if tuple(linear_merge1.shape[-2:]) != (oh, ow):      # _stem_hw restates the HRNet stem: fail loudly if it drifts
                raise RuntimeError('HRNetPN: the depth map was resized to %s but the RGB map is %s'
                                   % ((oh, ow), tuple(linear_merge1.shape[-2:])))
            return _feat1, _feat2, _feat3, f, {'merge1': merge1, 'merge2': _feat2,
                                               'linear_merge1': linear_merge1, 'linear_merge2': linear_merge2}
        return _feat1, _feat2, _feat3, avg1, avg2, avg3, f


NAME_TO_FUNC = {
    'RGBD2SHRNetSin': CMC3HRNetSGCNSingleHead,
    'RGBD2SHRNetPNSin': CMC3HRNetSGCNPN2SingleHead,
}


def register_model(key, ctor):
    """Plugin hook: add an encoder under ``modal+arch+('Sin'|'Mul')``."""
    NAME_TO_FUNC[key] = ctor


def _load_encoder(encoder, path, tag):
    print('Init {} from {}'.format(tag, path))
    ckpt = torch.load(path, map_location='cpu')
    own = encoder.state_dict()
    for k, v in ckpt.items():
        if k in own:
            own[k] = v
        else:
            print('{} not matched.'.format(k))
    encoder.load_state_dict(own)


def build_model(opt):
    key = opt.modal + opt.arch + ('Mul' if opt.jigsaw else 'Sin')
    if key not in NAME_TO_FUNC:
        raise NotImplementedError('model not supported: {}'.format(key))
    model = NAME_TO_FUNC[key](opt.arch, opt.head, opt.feat_dim, opt.in_channel_list, opt.linear_feat_map,
                              opt.width, opt.pool_method, opt)
    if getattr(opt, 'IN_Pretrain', None) is not None:
        if not opt.arch.startswith('HRNet'):
            raise NotImplementedError
        _load_encoder(model.encoder1, opt.IN_Pretrain, 'Encoder1')
    if getattr(opt, 'depth_Pretrain', None) is not None:
        if not opt.arch.startswith('HRNet'):
            raise NotImplementedError
        _load_encoder(model.encoder2, opt.depth_Pretrain, 'Encoder2')
    # MoCo's momentum copy: the reference builds it without `opt` and fails for the RGBD2S models
    # (build_backbone.py:561-562); here it gets the same options.
    model_ema = None
    if opt.mem == 'moco':
        model_ema = NAME_TO_FUNC[key](opt.arch, opt.head, opt.feat_dim, opt.in_channel_list,
                                      opt.linear_feat_map, opt.width, opt.pool_method, opt)
    return model, model_ema
