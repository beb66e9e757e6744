"""Assemble profiles/rNN_bank_pass_pmc*.json from the two rocprofv3 --pmc passes of tools/pmc_bank.py
(FETCH_SIZE and WRITE_SIZE in SEPARATE passes: they do not fit the 4 TCC slots together), applying the gfx950
read correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads -> x2).
usage: python tools/pmc_bank_json.py <out.json> <fetch_counter_csv> <write_counter_csv> <K> <n_data> <dtype>"""
import csv
import json
import sys

out, fcsv, wcsv, K, n, dtype = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]


def mean(path, counter):
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
         if 'bank_pass_' in r.get('Kernel_Name', '') and r['Counter_Name'] == counter]
    return sum(v) / len(v), len(v)


B, D = 32, 128
row = 2 if dtype == 'bf16' else 4
fetch, nf = mean(fcsv, 'FETCH_SIZE')
write, nw = mean(wcsv, 'WRITE_SIZE')
read_b, write_b = 2 * fetch * 1024, write * 1024
alg = B * (3 * (K + 1) * D * row + (K + 1) * 8 + 12 * D * 4)
res = {
    'kernel': 'bank_pass_kernel (fused gather pass of hcm_bank_nce_fused%s)' % ('_bf16' if dtype == 'bf16' else ''),
    'config': {'B': B, 'K': K, 'n_data': n, 'D': D, 'dtype': dtype},
    'command': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/pmc_bank.py   '
               '(and a separate pass with --pmc WRITE_SIZE); K=%d N=%d DTYPE=%s' % (K, n, dtype),
    'raw': {'FETCH_SIZE_KiB_mean': fetch, 'WRITE_SIZE_KiB_mean': write, 'launches': min(nf, nw)},
    'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> read bytes = '
                  '2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE * 1024 taken as is',
    'read_bytes_per_launch': int(read_b), 'write_bytes_per_launch': int(write_b),
    'traffic_bytes_per_launch': int(read_b + write_b), 'algorithmic_bytes_per_launch': alg,
    'traffic_over_algorithmic': round((read_b + write_b) / alg, 4),
}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
