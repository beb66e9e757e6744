"""Times the nine PointNet++ HIP ops at the shapes of BASELINE config 4 (HRNetPN, B=32, 4096 points,
256x256 maps): SA levels of Pointnet2MSG (networks/pointnet2_msg.py NPOINTS/RADIUS/NSAMPLE/MLPS), FP
levels and the pts2depth three_nn/three_interpolate (build_backbone.py:447-455).  torch.cuda.Event
timing (the ops launch on the current stream)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcmoco_amd.pointnet2_hip as pn

d = torch.device('cuda:0')
B = int(os.environ.get('B', 32))
torch.manual_seed(0)


VEC_PEAK_TFS = 157.3          # fp32 vector / matrix peak of the MI355X (guide); a distance evaluation = 3 sub + 3 mul-add + compare ~ 9 flop


def timeit(name, fn, reps=5, work=None, unit=''):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    extra = '' if work is None else '  %.1f %s' % (work / ms / 1e6, unit)
    if work is not None and 'evals/s' in unit:          # rows 12 / 13 / 16 against the fp32 vector peak
        extra += '  (%.3f of the fp32 vector peak at 9 flop per evaluation)' % (work * 9 / (ms * 1e-3) / (VEC_PEAK_TFS * 1e12))
    print('%-58s %9.3f ms%s' % (name, ms, extra), flush=True)
    return ms


def cloud(n):
    # points sampled WITH replacement from a depth surface (duplicates), like depth2pts
    base = torch.rand(B, n, 3, device=d) * torch.tensor([1.0, 2.0, 0.4], device=d)
    src = torch.randint(0, n, (B, n), device=d)
    return torch.gather(base, 1, src.unsqueeze(-1).expand(B, n, 3)).contiguous()


total = 0.0
levels = [(4096, 4096, (0.025, 0.125), (16, 32), 3), (4096, 1024, (0.125, 0.25), (16, 32), 96 + 3),
          (1024, 256, (0.25, 0.5), (16, 32), 256 + 3), (256, 64, (0.5, 1.0), (16, 32), 512 + 3)]
xyz = cloud(4096)
for n, m, radii, nsamples, cin in levels:
    temp = torch.empty(B, n, device=d)
    idx = torch.zeros(B, m, dtype=torch.int32, device=d)

    def fps():
        temp.fill_(1e10)
        pn.furthest_point_sampling_wrapper(B, n, m, xyz, temp, idx)
    total += timeit('FPS n=%d m=%d' % (n, m), fps, reps=3, work=B * n * m, unit='G dist-evals/s')
    xyz_t = xyz.transpose(1, 2).contiguous()
    new_t = torch.empty(B, 3, m, device=d)
    total += timeit('gather_points c=3 n=%d m=%d' % (n, m), lambda: pn.gather_points_wrapper(B, 3, n, m, xyz_t, idx, new_t))
    new_xyz = new_t.transpose(1, 2).contiguous()
    feats = torch.randn(B, cin, n, device=d)
    for r, ns in zip(radii, nsamples):
        bidx = torch.zeros(B, m, ns, dtype=torch.int32, device=d)
        total += timeit('ball_query n=%d m=%d r=%.3f ns=%d' % (n, m, r, ns),
                        lambda: pn.ball_query_wrapper(B, n, m, r, ns, new_xyz, xyz, bidx), work=B * n * m, unit='G pair-evals/s (upper)')
        out = torch.empty(B, cin, m, ns, device=d)
        by = out.numel() * 8 + bidx.numel() * 4
        total += timeit('group_points c=%d n=%d np=%d ns=%d' % (cin, n, m, ns),
                        lambda: pn.group_points_wrapper(B, cin, n, m, ns, feats, bidx, out), work=by, unit='GB/s')
        g = torch.zeros(B, cin, n, device=d)
        timeit('group_points_grad c=%d  [atomic, reference ABI]' % cin,
               lambda: pn.group_points_grad_wrapper(B, cin, n, m, ns, out, bidx, g), work=by, unit='GB/s')
        timeit('group_points_grad c=%d  [LDS float atomics, r02]' % cin,
               lambda: pn.scatter_add_lds(out.view(B, cin, m * ns), bidx.view(B, -1), None, n, 1), work=by, unit='GB/s')
        bflat = bidx.view(B, -1)
        timeit('   scatter plan (once per index tensor)', lambda: pn.scatter_plan(bflat, None, n, 1))
        total += timeit('group_points_grad c=%d  [planned, deterministic]' % cin,
                        lambda: pn.scatter_add_planned(out.view(B, cin, m * ns), bflat, None, n, 1), work=by, unit='GB/s')
    xyz = new_xyz

for n, m, c in [(256, 64, 1024), (1024, 256, 512), (4096, 1024, 512), (4096, 4096, 256), (65536, 4096, 128)]:
    unknown, known = cloud(n), cloud(m)
    dist2 = torch.empty(B, n, 3, device=d)
    idx = torch.empty(B, n, 3, dtype=torch.int32, device=d)
    total += timeit('three_nn n=%d m=%d' % (n, m), lambda: pn.three_nn_wrapper(B, n, m, unknown, known, dist2, idx),
                    reps=3, work=B * n * m, unit='G pair-evals/s')
    feats = torch.randn(B, c, m, device=d)
    w = torch.rand(B, n, 3, device=d)
    out = torch.empty(B, c, n, device=d)
    by = out.numel() * 4 * 4
    total += timeit('three_interpolate c=%d n=%d m=%d' % (c, n, m),
                    lambda: pn.three_interpolate_wrapper(B, c, m, n, feats, idx, w, out), work=by, unit='GB/s')
    g = torch.zeros(B, c, m, device=d)
    timeit('three_interpolate_grad c=%d  [atomic, reference ABI]' % c,
           lambda: pn.three_interpolate_grad_wrapper(B, c, n, m, out, idx, w, g), work=by, unit='GB/s')
    timeit('three_interpolate_grad c=%d  [LDS float atomics, r02]' % c,
           lambda: pn.scatter_add_lds(out, idx.view(B, -1), w, m, 3), work=by, unit='GB/s')
    iflat = idx.view(B, -1)
    timeit('   scatter plan (once per index tensor)', lambda: pn.scatter_plan(iflat, w, m, 3))
    total += timeit('three_interpolate_grad c=%d  [planned, deterministic]' % c,
                    lambda: pn.scatter_add_planned(out, iflat, w, m, 3), work=by, unit='GB/s')
    if n == 65536:
        # the empty-mask case of pts2depth (build_backbone.py:427-445): a quarter of the images send every pixel to points 0, 1, 2
        hub = idx.clone()
        hub[::4] = torch.arange(3, dtype=torch.int32, device=d)
        hflat = hub.view(B, -1)
        timeit('three_interpolate_grad c=%d, 25%% empty-mask images  [LDS float atomics, r02]' % c,
               lambda: pn.scatter_add_lds(out, hflat, w, m, 3), work=by, unit='GB/s')
        timeit('three_interpolate_grad c=%d, 25%% empty-mask images  [planned, deterministic]' % c,
               lambda: pn.scatter_add_planned(out, hflat, w, m, 3), work=by, unit='GB/s')
print('sum of one call each: %.2f ms' % total)
total = 0.0
# max over the ball (pointnet2_modules.py:60-63) at the four set-abstraction levels: [B, C_out, npoint, nsample]
for npoint, outs in [(4096, (32, 64)), (1024, (128, 128)), (256, (256, 256)), (64, (512, 512))]:
    for cout, ns in zip(outs, (16, 32)):
        x = torch.randn(B, cout, npoint, ns, device=d, requires_grad=True)
        y = pn.ball_max(x)
        gy = torch.randn_like(y)
        by = x.numel() * 4
        total += timeit('ball max forward  [%d, %d, %d, %d]' % (B, cout, npoint, ns), lambda: pn.ball_max(x), work=by, unit='GB/s')
        total += timeit('ball max backward [%d, %d, %d, %d]' % (B, cout, npoint, ns),
                        lambda: torch.autograd.grad(y, x, gy, retain_graph=True), work=by, unit='GB/s')
print('ball max, sum of one call each: %.2f ms' % total)
