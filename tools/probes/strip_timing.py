"""Per-phase times of the key-tile loop of strip_kernel<Dense> from the diagnostic library of tools/probes/strip_timing.sh
(s_memtime stamps of wave 0 of every workgroup; the counter runs at the shader clock: a kernel of 49 us spans 118 k counts).  usage: HCM_LIB=<diag .so> strip_timing.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hcmoco_amd import _lib
_lib.LIB_PATH = os.environ['HCM_LIB']
from hcmoco_amd import hip_ops

d = torch.device('cuda:0')
torch.manual_seed(0)
B, h, S, J = 32, 64, 400, 17
m1 = torch.randn(B, 128, h, h, device=d).contiguous(memory_format=torch.channels_last)
m2 = torch.randn(B, 128, h, h, device=d).contiguous(memory_format=torch.channels_last)
keep = torch.ones(B, dtype=torch.int32, device=d)
keep[::4] = 0
ind = torch.randint(0, h * h, (B, S), device=d)
pix = torch.randint(0, h * h, (B, J), device=d)
vis = torch.ones(B, J, dtype=torch.int32, device=d)


def run():
    hip_ops.fmap_losses(m1, m2, None, ind, keep, pix, vis, keep, None, 0.07, do_joint=False, gemm_dtype='fp32')


setbuf = C.CDLL(os.environ['HCM_LIB']).hcm_debug_strip_timing
setbuf.argtypes = [C.c_void_p]
for _ in range(3):
    run()
torch.cuda.synchronize()
nwg = ((S + 63) // 64) * B * 2
buf = torch.zeros(nwg * 2, 8, dtype=torch.int64, device=d)
assert setbuf(buf.data_ptr()) == 0
run()
torch.cuda.synchronize()
setbuf(None)
t = buf.cpu().view(nwg, 2, 8).float()
names = ['prologue (Q fragments, first tile, first GEMM issued)', 'accumulator read + commit of the next tile (waits: GEMM result, key registers)',
         'barrier', 'fetch issue + operand reads + MFMA issue', 'element-wise code (+ GEMM 2 in the grad pass)']
for p, tag in ((0, 'stats pass'), (1, 'grad pass')):
    x = t[:, p]
    live = x[:, 5] > 0
    x = x[live]
    tiles = x[:, 5].mean()
    x[:, 5] = x[:, 5].clamp(min=1)
    print('%s: %d workgroups stamped, %.0f tiles each; shader-clock cycles per workgroup (wave 0), mean / max over workgroups' % (tag, x.shape[0], tiles))
    tot = x[:, :5].sum(1)
    for k in range(5):
        per = x[:, k] / (x[:, 5] if k else 1)
        print('   %-86s %8.1f / %8.1f  %s' % (names[k], x[:, k].mean(), x[:, k].max(), '' if k == 0 else '(%.1f per tile)' % per.mean()))
    print('   %-86s %8.1f / %8.1f' % ('sum', tot.mean(), tot.max()))
