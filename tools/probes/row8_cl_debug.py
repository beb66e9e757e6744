import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from hcmoco_amd import hip_ops as ops
from test_section_gpu import make_maps
d = torch.device('cuda:0')
for (B, width, size, R) in [(3, 18, 16, 37), (2, 18, 64, 100)]:
    torch.manual_seed(1)
    m1, m2 = make_maps(B, width, size, 3), make_maps(B, width, size, 4)
    Ctot = 15 * width
    Wp = [torch.randn(128, Ctot, 1, 1) * 0.05 for _ in range(2)]
    bp = [torch.randn(128) * 0.1 for _ in range(2)]
    pix = torch.randint(0, size * size, (B, R))
    g = lambda t: t.to(d)
    gm1, gm2 = [g(t) for t in m1], [g(t) for t in m2]
    a = ops.project_rows(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]), channels_last=True)
    b = ops.project_rows(gm1, gm2, g(pix), g(Wp[0]), g(bp[0]), g(Wp[1]), g(bp[1]), channels_last=False)
    torch.cuda.synchronize()
    xa, xb = a[1].cpu(), b[1].cpu()
    print((B, width, size, R), 'rows equal', torch.equal(a[0], b[0]), 'xs equal', torch.equal(xa, xb))
    bad = (xa != xb)
    print('  bad per modality', bad.flatten(1).sum(1).tolist(), 'of', xa[0].numel())
    cols = bad.any(1)   # [2, ld]
    for m in range(2):
        print('  modality', m, 'bad columns', cols[m].nonzero().flatten().tolist()[:40])
        rowsbad = bad[m].any(1).nonzero().flatten().tolist()
        print('  bad rows', rowsbad[:20], 'n', len(rowsbad))

# ---- the channels-last workspace itself against torch.permute
import ctypes as C
from hcmoco_amd import _lib
from hcmoco_amd.hip_ops import _branches, _dev, _stream
L = _lib.lib()
B, width, size, R = 3, 18, 16, 37
torch.manual_seed(1)
m1, m2 = make_maps(B, width, size, 3), make_maps(B, width, size, 4)
Ctot = 15 * width
g = lambda t: t.to(d)
gm1, gm2 = [g(t) for t in m1], [g(t) for t in m2]
Wp = [g(torch.randn(128, Ctot) * 0.05) for _ in range(2)]
bp = [g(torch.randn(128) * 0.1) for _ in range(2)]
pix = g(torch.randint(0, size * size, (B, R)))
br1, br2 = _branches(gm1, 'x'), _branches(gm2, 'x')
n = int(L.hcm_project_rows_nhwc_floats(br1, br2, B, Ctot))
ws = torch.full((n,), -7.0, device=d)
rows = torch.empty(2, B * R, 128, device=d)
p = lambda t: C.c_void_p(t.data_ptr())
rc = L.hcm_project_rows_cl(br1, br2, B, p(pix), R, Ctot, 128, p(Wp[0]), p(bp[0]), p(Wp[1]), p(bp[1]), C.c_void_p(0), p(rows),
                           C.c_void_p(0), p(ws), n, _stream())
torch.cuda.synchronize()
print('rc', rc, 'floats', n, 'untouched', int((ws == -7.0).sum()))
off = 0
for m, maps in enumerate((gm1, gm2)):
    for i in range(2):
        t = maps[i]
        ref = t.permute(0, 2, 3, 1).reshape(-1)
        got = ws[off:off + ref.numel()]
        print('modality', m, 'branch', i, 'shape', tuple(t.shape), 'equal', torch.equal(got, ref), 'offset', off)
        off += ref.numel()
