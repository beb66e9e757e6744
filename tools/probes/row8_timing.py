"""Per-phase cycle counts inside project_rows_kernel / branch_grad_t_kernel from the diagnostic library built by
tools/probes/row8_timing.sh (s_memtime stamps, 16 slots per workgroup).  usage: HCM_LIB=<diag .so> row8_timing.py [crop] [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd import _lib
_lib.LIB_PATH = os.environ['HCM_LIB']
from hcmoco_amd import hip_ops

crop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
d = torch.device('cuda:0')
torch.manual_seed(0)
width, S, J, F = 18, 400, 17, 128
h = crop // 4
maps1 = [torch.randn(B, width * 2 ** i, h >> i, h >> i, device=d) for i in range(4)]
maps2 = [torch.randn(B, width * 2 ** i, h >> i, h >> i, device=d) for i in range(4)]
Ctot = 15 * width
Wp = [torch.randn(F, Ctot, 1, 1, device=d) * 0.05 for _ in range(2)]
bp = [torch.randn(F, device=d) * 0.1 for _ in range(2)]
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
setbuf = C.CDLL(os.environ['HCM_LIB']).hcm_debug_row8_timing
setbuf.argtypes = [C.c_void_p]
try:
    print('hipOccupancyMaxActiveBlocksPerMultiprocessor(branch_grad_t_kernel, 256 threads) =', C.CDLL(os.environ['HCM_LIB']).hcm_debug_row8_occupancy())
except AttributeError:
    pass
buf = torch.zeros(1 << 20, 16, dtype=torch.int64, device=d)


def run(fn, name, nslots):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    assert setbuf(buf.data_ptr()) == 0
    fn()
    torch.cuda.synchronize()
    setbuf(None)
    t = buf.cpu()
    used = t[:, 0] != 0
    global wg_index
    wg_index = used.nonzero().view(-1)
    t = t[used]
    print('%s: %d workgroups stamped' % (name, t.shape[0]))
    t0 = t[:, 0].min()
    print('  kernel span (first start -> last stamp) %d s_memtime counts (shader clock: ~2.4 counts per ns)' % int((t.max() - t0)))
    return t, t0


rows_, xs_, _ = hip_ops.project_rows(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1])
t, t0 = run(lambda: hip_ops.project_rows(maps1, maps2, pix, Wp[0], bp[0], Wp[1], bp[1]), 'project_rows_kernel', 12)
st = (t[:, 1] - t[:, 0]).float()
print('  start -> maps staged + B fragments: mean %.0f max %.0f' % (st.mean(), st.max()))
print('    of which: staging loops issued %.0f, B fragment loads issued %.0f, to the first barrier passed %.0f (means, wave 0)' % ((t[:, 12] - t[:, 0]).float().mean(), (t[:, 13] - t[:, 12]).float().mean(), (t[:, 1] - t[:, 13]).float().mean()))
for k in range(5):
    ok = t[:, 3 + 2 * k] != 0
    if not bool(ok.any()):
        break
    prev = t[:, 1] if k == 0 else t[:, 1 + 2 * k]
    ga = (t[ok, 2 + 2 * k] - prev[ok]).float()
    mf = (t[ok, 3 + 2 * k] - t[ok, 2 + 2 * k]).float()
    print('  tile %d: gather mean %.0f max %.0f   multiply+store mean %.0f max %.0f   (%d workgroups)' % (k, ga.mean(), ga.max(), mf.mean(), mf.max(), int(ok.sum())))
life = (t.max(dim=1).values - t[:, 0]).float()
print('  workgroup lifetime mean %.0f max %.0f; start offsets: mean %.0f max %.0f' % (life.mean(), life.max(), (t[:, 0] - t0).float().mean(), (t[:, 0] - t0).float().max()))

gr = torch.randn(2, B * (S + J), F, device=d)
gr.view(2, B, S + J, F)[:, keep == 0, :S] = 0
bw = lambda: hip_ops.project_rows_backward(gr, xs_, Wp[0], Wp[1], dpooled, scale, pix, shapes, keep, S, False)
t, t0 = run(bw, 'branch_grad_t_kernel', 5)
empty = t[:, 4] != 0
full = t[:, 3] != 0
print('  workgroups: %d with entries, %d empty' % (int(full.sum()), int(empty.sum())))
e = t[empty]
print('  empty: start -> offsets %.0f, -> stored %.0f (mean)' % ((e[:, 1] - e[:, 0]).float().mean(), (e[:, 4] - e[:, 1]).float().mean()))
f = t[full]
for name, a, b in (('entry -> tiles()', 5, 0), ('offsets issued', 0, 6), ('offsets landed', 6, 1), ('offsets', 0, 1), ('T accumulate', 1, 2), ('multiply + store', 2, 3)):
    v = (f[:, b] - f[:, a]).float()
    print('  with entries: %-18s mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f' % (name, v.mean(), v.median(), v.quantile(0.9), v.max()))
life = (t.max(dim=1).values - t[:, 0]).float()
print('  lifetime mean %.0f max %.0f; start offsets mean %.0f max %.0f (span %.0f)' % (life.mean(), life.max(), (t[:, 0] - t0).float().mean(), (t[:, 0] - t0).float().max(), float(t.max() - t0)))
# how many workgroups are alive over time (10 bins)
span = float(t.max() - t0)
starts, ends = (t[:, 0] - t0).float(), (t.max(dim=1).values - t0).float()
print('  alive workgroups at 10 %% steps:', [int(((starts <= x * span / 10) & (ends > x * span / 10)).sum()) for x in range(10)])

# per branch: grid (B, 2, tiles), tiles in dispatch order (hcm_project_rows_backward: most expensive kind first; at 256 x 256
# with HRNet-w18 that is branch 3, 2, 1, 0 (rows-first path: cheapest) with 4 / 16 / 16 / 16 workgroups per image and modality)
if crop == 256:
    order, cnt = [3, 2, 1, 0], [4, 16, 16, 16]
    z = wg_index // (2 * B)
    lo = 0
    for i, c in zip(order, cnt):
        sel = (z >= lo) & (z < lo + c)
        lo += c
        tt = t[sel]
        fl = tt[:, 3] != 0
        lf = (tt.max(dim=1).values - tt[:, 0]).float()
        msg = '  branch %d: %4d workgroups (%4d with entries) lifetime mean %6.0f max %6.0f' % (i, int(sel.sum()), int(fl.sum()), lf.mean(), lf.max())
        if bool(fl.any()):
            ff = tt[fl]
            msg += '   offsets %6.0f  T %6.0f  multiply %6.0f (means, with entries)' % ((ff[:, 1] - ff[:, 0]).float().mean(), (ff[:, 2] - ff[:, 1]).float().mean(), (ff[:, 3] - ff[:, 2]).float().mean())
        print(msg)
