"""Stand-alone launches of the dense soft-InfoNCE + SCL kernels at the bench size (B=32, 64x64 maps, S=400, J=17),
for rocprofv3 --pmc passes (MFMA counters of strip_kernel).  FMAP_DTYPE=bf16 selects the bf16 contractions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcmoco_amd import _lib, hip_ops
if os.environ.get('HCM_LIB'):                 # a diagnostic build of the library
    _lib.LIB_PATH = os.environ['HCM_LIB']

d = torch.device('cuda:0')
torch.manual_seed(0)
B, h, S, J = 32, 64, 400, 17
m1 = torch.randn(B, 128, h, h, device=d).contiguous(memory_format=torch.channels_last)
m2 = torch.randn(B, 128, h, h, device=d).contiguous(memory_format=torch.channels_last)
keep = torch.ones(B, dtype=torch.int32, device=d)
keep[::4] = 0                                              # 24 of 32 images carry depth, as in the bench batches
ind = torch.randint(0, h * h, (B, S), device=d)
pix = torch.randint(0, h * h, (B, J), device=d)
vis = torch.ones(B, J, dtype=torch.int32, device=d)
for i in range(6):
    hip_ops.fmap_losses(m1, m2, None, ind, keep, pix, vis, keep, None, 0.07, do_joint=False,
                        gemm_dtype=os.environ.get('FMAP_DTYPE', 'fp32'))
torch.cuda.synchronize()
print('done')
