"""Micro-benchmark of the fused bank-NCE pass on the GPU box (hipEvent timing through the C ABI).
Prints achieved algorithmic GB/s:  bytes = 3*B*(K+1)*D*4 + B*(K+1)*8 + 12*B*D*4  (SURVEY 8d)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcmoco_amd import hip_ops

d = torch.device('cuda:0')
torch.manual_seed(0)
nrm = torch.nn.functional.normalize
for n, K, B in [(131072, 16384, 32), (131072, 65536, 32), (1048576, 65536, 32), (131072, 4096, 32)]:
    D = 128
    banks = [nrm(torch.randn(n, D, device=d)) for _ in range(3)]
    xs = [nrm(torch.randn(B, D, device=d)) for _ in range(3)]
    idx = torch.randint(0, n, (B, K + 1), device=d)
    hip_ops.bank_nce_fused_timed(banks, idx, xs, 0.07, 3)
    ms = hip_ops.bank_nce_fused_timed(banks, idx, xs, 0.07, 20)
    by = 3 * B * (K + 1) * D * 4 + B * (K + 1) * 8 + 12 * B * D * 4
    print('n=%d K=%d B=%d  %.3f ms/pass  %.1f GB/s algorithmic (%.1f MB)' % (n, K, B, ms, by / ms / 1e6, by / 1e6), flush=True)
