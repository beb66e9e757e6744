"""The bank pass over the regimes that matter: n_data x K x storage type x kernel variant.

    python tools/bank_sweep.py time OUT.json      every configuration in its own subprocess (the library's state is read
                                                  once per process): mean duration of the PASS kernel (hipEvents around
                                                  it, hcm_prof_*) back to back, and 'cold' -- a 1 GiB fill between two
                                                  launches pushes the banks out of the 256 MiB Infinity Cache, which is
                                                  what the encoders' activations do between two training steps
    python tools/bank_sweep.py worker ...         (internal)
    python tools/bank_sweep.py pmc N K DTYPE      six plain launches for a rocprofv3 --pmc pass (tools/run_bank_sweep.sh)

Algorithmic bytes per launch: B * (3 (K+1) D s + (K+1) 8 + 12 D 4), s = 4 (fp32) / 2 (bf16)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, D = 32, 128
NS = (131072, 1048576, 4194304)
KS = (16384, 65536)


def alg_bytes(K, dtype):
    return B * (3 * (K + 1) * D * (2 if dtype == 'bf16' else 4) + (K + 1) * 8 + 12 * D * 4)


def setup(n, K, dtype):
    import torch
    d = torch.device('cuda:0')
    torch.manual_seed(0)
    banks = []
    for _ in range(3):
        b = torch.empty(n, D, device=d)
        for lo in range(0, n, 262144):                   # in pieces: no 2 GB temporaries
            b[lo:lo + 262144] = torch.nn.functional.normalize(torch.randn(min(262144, n - lo), D, device=d))
        banks.append(b.to(torch.bfloat16) if dtype == 'bf16' else b)
    xs = [torch.nn.functional.normalize(torch.randn(B, D, device=d)) for _ in range(3)]
    idxs = [torch.randint(0, n, (B, K + 1), device=d) for _ in range(4)]
    return banks, xs, idxs


def worker(n, K, dtype):
    sys.path.insert(0, ROOT)
    import torch
    from hcmoco_amd import hip_ops
    banks, xs, idxs = setup(n, K, dtype)
    flush = torch.empty(2 ** 28, dtype=torch.float32, device='cuda:0')          # 1 GiB
    out = {}
    for mode in ('back_to_back', 'cold'):
        for _ in range(3):
            hip_ops.bank_nce_fused_raw(banks, idxs[0], xs, 0.07)
        torch.cuda.synchronize()
        hip_ops.prof_enable(True)
        for i in range(12):
            if mode == 'cold':
                flush.fill_(float(i))
            hip_ops.bank_nce_fused_raw(banks, idxs[i % 4], xs, 0.07)
        ms, cnt = hip_ops.prof_read('bank_pass')
        hip_ops.prof_enable(False)
        out[mode + '_us'] = round(1e3 * ms / cnt, 2)
        out[mode + '_GBps'] = round(alg_bytes(K, dtype) / (ms / cnt * 1e-3) / 1e9, 1)
    print(json.dumps(out))


def pmc(n, K, dtype):
    sys.path.insert(0, ROOT)
    import torch
    from hcmoco_amd import hip_ops
    banks, xs, idxs = setup(n, K, dtype)
    for i in range(6):
        hip_ops.bank_nce_fused_raw(banks, idxs[i % 4], xs, 0.07)
    torch.cuda.synchronize()


def main():
    if sys.argv[1] == 'worker':
        return worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    if sys.argv[1] == 'pmc':
        return pmc(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    rows = []
    for n in NS:
        for K in KS:
            for dtype in ('fp32', 'bf16'):
                # r05: ONE kernel per dtype is left (csrc/bank_lean.hip, ring 2 fp32 / ring 4 bf16, 512 rows per workgroup); the
                # variants r03 / r04 swept here (general kernel rings, LDS-DMA rings, 256 rows) are in profiles/r0{3,4}_bank_pass_sweep.json
                for name in ({'fp32': 'lean2_r512', 'bf16': 'lean4_r512'}[dtype],):
                    env = dict(os.environ)
                    res = subprocess.run([sys.executable, os.path.abspath(__file__), 'worker', str(n), str(K), dtype],
                                         capture_output=True, text=True, env=env, timeout=600)
                    line = [l for l in res.stdout.splitlines() if l.startswith('{')]
                    row = {'n_data': n, 'K': K, 'dtype': dtype, 'variant': name, 'bank_MB_total': round(3 * n * D * (2 if dtype == 'bf16' else 4) / 1e6),
                           'algorithmic_bytes': alg_bytes(K, dtype)}
                    row.update(json.loads(line[-1]) if line else {'error': res.stderr.strip().splitlines()[-1:]})
                    rows.append(row)
                    print(json.dumps(row), flush=True)
    json.dump(rows, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
