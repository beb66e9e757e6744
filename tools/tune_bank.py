"""Micro-benchmark of the fused bank-NCE pass on the GPU box (hipEvent timing through the C ABI).
Sweeps the kernel variants (HCM_BANK_VARIANT / HCM_BANK_ROWS are read once per process, so each
configuration runs in its own subprocess).  Prints achieved algorithmic GB/s:
bytes = 3*B*(K+1)*D*4 + B*(K+1)*8 + 12*B*D*4  (SURVEY 8d)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'worker':
    sys.path.insert(0, ROOT)
    import torch
    from hcmoco_amd import hip_ops
    d = torch.device('cuda:0')
    torch.manual_seed(0)
    nrm = torch.nn.functional.normalize
    for n, K, B in [(131072, 16384, 32), (131072, 131072, 32)]:
        D = 128
        banks = [nrm(torch.randn(n, D, device=d)) for _ in range(3)]
        xs = [nrm(torch.randn(B, D, device=d)) for _ in range(3)]
        idx = torch.randint(0, n, (B, K + 1), device=d)
        for name, bk, sz in (('fp32', banks, 4), ('bf16', [b.bfloat16() for b in banks], 2)):
            hip_ops.bank_nce_fused_timed(bk, idx, xs, 0.07, 3)
            whole = hip_ops.bank_nce_fused_timed(bk, idx, xs, 0.07, 20)
            by = 3 * B * (K + 1) * D * sz + B * (K + 1) * 8 + 12 * B * D * 4
            print('  K=%6d %s whole op %.1f us  (%.0f GB/s algorithmic over the whole op)' % (K, name, 1e3 * whole, by / whole / 1e6),
                  flush=True)
else:
    for variant in (1, 2, 3, 4, 6):          # = prefetch depth NPF
        for rows in (0,):
            env = dict(os.environ, HCM_BANK_VARIANT=str(variant), HCM_BANK_ROWS=str(rows))
            print('variant %d rows %d' % (variant, rows), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), 'worker'], env=env)
