"""Stand-alone launches of hcm_conv3x3_wgrad for rocprofv3 --pmc passes (SHAPE=N,C,H)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcmoco_amd import hip_ops
N, C, H = (int(v) for v in os.environ.get('SHAPE', '32,18,64').split(','))
x = torch.randn(N, C, H, H, device='cuda'); dy = torch.randn(N, C, H, H, device='cuda')
for _ in range(6):
    hip_ops.conv3x3_wgrad(x, dy)
torch.cuda.synchronize()
