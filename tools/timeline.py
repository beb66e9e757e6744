"""Timeline statistics of ONE steady-state step from a rocprofv3 --kernel-trace CSV (run on the GPU box):
time with 0 / 1 / 2 / 3+ kernels running, the largest all-idle gaps with the kernels around them, and busy time per
queue.  usage: python tools/timeline.py <kernel_trace.csv> [marker-substring]"""
import csv
import sys
from collections import defaultdict

trace = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'bank_pass_kernel'
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
lo, hi = marks[-2], marks[-1]
win = rows[lo:hi]
t0, t1 = win[0][0], rows[hi][0]
ev = []
for s, e, _, _ in win:
    ev.append((s, 1))
    ev.append((min(e, t1), -1))
ev.sort()
hist = defaultdict(int)
cur, last = 0, t0
gaps = []
for t, d in ev:
    if t > last:
        hist[min(cur, 3)] += t - last
        if cur == 0:
            gaps.append((t - last, last))
    cur += d
    last = t
tot = t1 - t0
print('window %.2f ms' % (tot / 1e6))
for k in sorted(hist):
    print('  %s kernels running: %6.2f ms (%.1f %%)' % (k if k < 3 else '3+', hist[k] / 1e6, 100.0 * hist[k] / tot))
perq = defaultdict(int)
for s, e, _, q in win:
    perq[q] += e - s
print('busy per queue:', {q: round(v / 1e6, 2) for q, v in sorted(perq.items(), key=lambda kv: -kv[1])})
gaps.sort(reverse=True)
print('idle gaps > 15 us: %d, total %.2f ms; all gaps total %.2f ms' % (sum(1 for g in gaps if g[0] > 15000),
                                                                        sum(g[0] for g in gaps if g[0] > 15000) / 1e6,
                                                                        sum(g[0] for g in gaps) / 1e6))
for g, at in gaps[:12]:
    before = max((r for r in win if r[1] <= at + 1), key=lambda r: r[1], default=None)
    after = min((r for r in win if r[0] >= at + g - 1), key=lambda r: r[0], default=None)
    print('  %.1f us at +%.2f ms  after %s | before %s' % (g / 1e3, (at - t0) / 1e6, before[2][:50] if before else '-',
                                                           after[2][:50] if after else '-'))
