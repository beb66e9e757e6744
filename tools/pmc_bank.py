"""Stand-alone launches of the fused bank-NCE op at the bench size, for rocprofv3 --pmc passes:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_bank.py
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_bank.py
(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; MI355X_MICROARCH.md)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcmoco_amd import hip_ops

d = torch.device('cuda:0')
torch.manual_seed(0)
B, K, n, D = 32, int(os.environ.get('K', 16384)), int(os.environ.get('N', 131072)), 128
nrm = torch.nn.functional.normalize
banks = [nrm(torch.randn(n, D, device=d)) for _ in range(3)]
if os.environ.get('DTYPE', 'f32') == 'bf16':          # BASELINE config 5 bank storage
    banks = [b.to(torch.bfloat16) for b in banks]
xs = [nrm(torch.randn(B, D, device=d)) for _ in range(3)]
for i in range(6):
    idx = torch.randint(0, n, (B, K + 1), device=d)
    hip_ops.bank_nce_fused_raw(banks, idx, xs, 0.07)
torch.cuda.synchronize()
print('done')
