"""The serial part of a step in a rocprofv3 kernel trace: every launch between the last forward kernel of the encoders
and the first backward kernel of the encoders (the heads, the projections, the feature-map and bank losses and their
backward), with start offsets, durations and the idle gap before each.  usage: loss_section.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'bank_pass_' in r[2]]
m = marks[-2]
# walk back to the last encoder-forward kernel (bn_apply / conv / winograd) before the marker, forward to the first bn_bwd
enc_fwd = ('bn_apply_kernel', 'bn_small_fwd', 'conv3x3_mfma', 'miopenSp3AsmConv', 'bn_stats_kernel', 'upsample_bilinear_kernel',
           'sgc_mix_kernel', 'sgc_norm_kernel')        # (the SemGCN is the third encoder; since r03 it runs on the caller's stream)
lo = m
while lo > 0 and not any(k in rows[lo][2] for k in enc_fwd):
    lo -= 1
hi = m
while hi < len(rows) and 'bn_bwd' not in rows[hi][2] and 'bn_small_bwd' not in rows[hi][2] and 'sgc_bwd' not in rows[hi][2]:
    hi += 1
t0 = rows[lo][1]
print('loss section: %d launches, %.3f ms from the end of the last encoder forward kernel to the first encoder backward kernel'
      % (hi - lo - 1, (rows[hi][0] - t0) / 1e6))
busy = 0
last_end = t0
gaps = 0
for s, e, n, q in rows[lo + 1:hi]:
    gap = max(0, s - last_end)
    gaps += gap
    busy += e - s
    print('+%8.1f us  gap %6.1f  dur %6.1f  q%s  %s' % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, q, n[:90]))
    last_end = max(last_end, e)
print('busy %.3f ms, idle gaps %.3f ms' % (busy / 1e6, gaps / 1e6))
