"""The pts2depth backward (three_interpolate_grad: B=32, c=128, 65536 pixels -> 4096 points) stand-alone, for rocprofv3:
random neighbours, then with a quarter of the images sending every pixel to points 0, 1, 2 (empty masks)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hcmoco_amd.pointnet2_hip as pn

d = torch.device('cuda:0')
B, c, n, m = 32, 128, 65536, 4096
torch.manual_seed(0)
mode = sys.argv[1] if len(sys.argv) > 1 else 'random'
if mode == 'nn':                                   # real nearest neighbours of a pixel grid: neighbouring pixels share targets
    ys, xs = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing='ij')
    unknown = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.zeros(n)], 1).to(d).expand(B, n, 3).contiguous()
    sel = torch.stack([torch.randperm(n // 4)[:m] for _ in range(B)]).to(d)       # points inside the top quarter of the image
    known = torch.gather(unknown, 1, sel.unsqueeze(-1).expand(B, m, 3)).contiguous()
    dist2 = torch.empty(B, n, 3, device=d)
    idx = torch.empty(B, n, 3, dtype=torch.int32, device=d)
    pn.three_nn_wrapper(B, n, m, unknown, known, dist2, idx)
else:
    idx = torch.randint(0, m, (B, n, 3), dtype=torch.int32, device=d)
    if mode == 'hub':
        idx[::4] = torch.arange(3, dtype=torch.int32, device=d)
w = None if os.environ.get('NOW') else torch.rand(B, n, 3, device=d)
g = torch.randn(B, c, n, device=d)
iflat = idx.view(B, -1)
for _ in range(3):
    out = pn.scatter_add_planned(g, iflat, w, m, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    out = pn.scatter_add_planned(g, iflat, w, m, 3)
e1.record()
torch.cuda.synchronize()
print(mode, '%.3f ms' % (e0.elapsed_time(e1) / 5))
