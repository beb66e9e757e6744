"""Row 8 of the loss section (merge_all_res + 1x1 projection at the sampled pixels, networks/build_backbone.py:243-254)
stand-alone at the bench size, for hipEvent timings and rocprofv3 --pmc passes: sample/projection forward, the two
backward contractions and branch_grad, plus variants of the inputs that separate the phases of branch_grad
(all rows dropped -> list building only; R = J -> almost nothing listed).
usage: row8_probe.py [crop=256] [B=32] [reps=20]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd import hip_ops
hip_ops.ROW8_CHANNELS_LAST = os.environ.get('ROW8_CL', '0') == '1'      # r06: hcm_project_rows_cl under the counters

crop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = torch.device('cuda:0')
torch.manual_seed(0)
width, S, J, F = 18, 400, 17, 128
h = crop // 4
maps1 = [torch.randn(B, width * 2 ** i, h >> i, h >> i, device=d) for i in range(4)]
maps2 = [torch.randn(B, width * 2 ** i, h >> i, h >> i, device=d) for i in range(4)]
Ctot = 15 * width
Wp = [torch.randn(F, Ctot, 1, 1, device=d) * 0.05 for _ in range(2)]
bp = [torch.randn(F, device=d) * 0.1 for _ in range(2)]
# pixels the way the bench batches have them: a centred disc of radius 3/8 of the map, 15 % of the joints on pixel 0
yy, xx = torch.meshgrid(torch.arange(h), torch.arange(h), indexing='ij')
disc = (((yy - h / 2) ** 2 + (xx - h / 2) ** 2) <= (0.375 * h) ** 2).reshape(-1).nonzero().view(-1)
pix = disc[torch.randint(0, disc.numel(), (B, S + J))].to(d)
pix[:, S:][torch.rand(B, J, device=d) < 0.15] = 0
keep = torch.ones(B, dtype=torch.int32, device=d)
keep[::4] = 0
pix[keep == 0, :S] = 0
dpooled = torch.randn(2, B, Ctot, device=d)
scale = torch.tensor(1.0, device=d)
shapes = [tuple(m.shape) for m in maps1]


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%-52s %8.1f us' % (name, 1e3 * e0.elapsed_time(e1) / reps))


fwd = getattr(hip_ops, 'project_rows', None)
if fwd is not None:
    timed('project_rows (sample + projection, MFMA)', lambda: fwd(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1]))
if hasattr(hip_ops, 'sample_branches'):
    xs, Wpad, grows = hip_ops.sample_branches(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1])
    timed('sample_branches (xs staging)', lambda: hip_ops.sample_branches(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1]))
    timed('bmm rows = xs Wpad^T (rocBLAS)', lambda: torch.bmm(xs, Wpad.transpose(1, 2)))
    grows = torch.randn_like(grows)
    timed('bmm dxs = grows Wpad (rocBLAS)', lambda: torch.bmm(grows, Wpad))
    timed('bmm dWpad = grows^T xs (rocBLAS)', lambda: torch.bmm(grows.transpose(1, 2), xs))
    dxs = torch.bmm(grows, Wpad)
    dWpad = torch.bmm(grows.transpose(1, 2), xs)
    timed('branch_grad (bench-like pixels)', lambda: hip_ops.branch_grad(dxs, dpooled, scale, pix, shapes, dWpad, F, keep, S))
    allkeep = torch.ones_like(keep)
    timed('branch_grad (no image dropped)', lambda: hip_ops.branch_grad(dxs, dpooled, scale, pix, shapes, dWpad, F, allkeep, S))
    nokeep = torch.zeros_like(keep)
    timed('branch_grad (every image dropped: J rows listed)', lambda: hip_ops.branch_grad(dxs, dpooled, scale, pix, shapes, dWpad, F, nokeep, S))
    pj = pix[:, S:].contiguous()
    dxj = dxs.view(2, B, S + J, -1)[:, :, S:].contiguous().view(2, B * J, -1)
    timed('branch_grad (R = J rows only)', lambda: hip_ops.branch_grad(dxj, dpooled, scale, pj, shapes, None, F, None, 0))
    nopix = torch.empty(B, 0, dtype=torch.int64, device=d)
    timed('branch_grad (R = 0: pooling gradient only)', lambda: hip_ops.branch_grad(None, dpooled, None, nopix, shapes))
bwd = getattr(hip_ops, 'project_rows_backward', None)
if bwd is not None and fwd is not None:
    rows_, xs_, _ = fwd(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1])
    gr = torch.randn(2, B * (S + J), F, device=d)
    gr.view(2, B, S + J, F)[:, keep == 0, :S] = 0
    timed('project_rows_backward (plan + dW + branch tiles)',
          lambda: bwd(gr, xs_, Wp[0], Wp[1], dpooled, scale, pix, shapes, keep, S))
    timed('project_rows_backward (no weight gradients)',
          lambda: bwd(gr, xs_, Wp[0], Wp[1], dpooled, scale, pix, shapes, keep, S, False))
    timed('project_rows_backward (no image dropped: hub pixel 0)',
          lambda: bwd(gr, xs_, Wp[0], Wp[1], dpooled, scale, pix, shapes, None, 0))
    pj = pix[:, S:].contiguous()
    rj, xj, _ = fwd(maps1, maps2, pj, Wp[0], bp[0], Wp[1], bp[1])
    timed('project_rows_backward (R = J rows only)',
          lambda: bwd(gr[:, :B * J].contiguous(), xj, Wp[0], Wp[1], dpooled, scale, pj, shapes, None, 0))
print('done')
