"""Timeline statistics of ONE steady-state step from a rocprofv3 --kernel-trace CSV (run on the GPU box):
time with 0 / 1 / 2 / 3+ kernels running, the largest all-idle gaps with the kernels around them, and busy time per
queue.  usage: python tools/timeline.py <kernel_trace.csv> [marker-substring]"""
import csv
import sys
from collections import defaultdict

trace = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'bank_pass_'
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

# time each kernel name spends as the ONLY kernel on the GPU (the serial part of the step), and a coarse
# picture of the step: per 1-ms slot, the number of queues that ran anything
alone = defaultdict(int)
calls = defaultdict(int)
active = {}
last = t0
ev2 = []
for i, (s, e, n, q) in enumerate(win):
    ev2.append((s, 1, i))
    ev2.append((min(e, t1), -1, i))
ev2.sort(key=lambda x: (x[0], x[1]))
for t, dlt, i in ev2:
    if t > last and len(active) == 1:
        alone[win[next(iter(active))][2]] += t - last
    if dlt > 0:
        active[i] = True
        calls[win[i][2]] += 1
    else:
        active.pop(i, None)
    last = t
print('time as the only running kernel (top 30 of %.2f ms):' % (sum(alone.values()) / 1e6))
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:30]:
    print('  %8.1f us  %4d calls  %s' % (v / 1e3, calls[n], n[:110]))

# the serial section between the encoders' forward and backward: the longest stretches in which the two busiest
# queues (the encoder streams) run nothing, with what the other queues ran meanwhile
top2 = [q for q, _ in sorted(perq.items(), key=lambda kv: -kv[1])[:2]]
enc = sorted((s, min(e, t1)) for s, e, _, q in win if q in top2)
holes, cur_end = [], t0
for s, e in enc:
    if s > cur_end:
        holes.append((s - cur_end, cur_end, s))
    cur_end = max(cur_end, e)
holes.append((t1 - cur_end, cur_end, t1))
holes.sort(reverse=True)
for length, a, b in holes[:3]:
    inside = [(s, e, n) for s, e, n, q in win if q not in top2 and s >= a and e <= b]
    busy = sum(e - s for s, e, _ in inside)
    print('encoder streams both idle for %.2f ms at +%.2f ms: %d kernels on other queues, %.2f ms busy' %
          (length / 1e6, (a - t0) / 1e6, len(inside), busy / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in inside:
        agg[n][0] += e - s
        agg[n][1] += 1
    for n, (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print('    %7.1f us %3d x  %s' % (v / 1e3, c, n[:100]))

# the long kernels: everything above 40 us in the window, by name
long_ = defaultdict(lambda: [0, 0])
for s, e, n, q in win:
    if e - s > 40000:
        long_[n][0] += e - s
        long_[n][1] += 1
print('kernels longer than 40 us: %.2f ms in %d launches' % (sum(v[0] for v in long_.values()) / 1e6, sum(v[1] for v in long_.values())))
for n, (v, c) in sorted(long_.items(), key=lambda kv: -kv[1][0])[:25]:
    print('  %8.1f us  %4d x %6.1f us  %s' % (v / 1e3, c, v / c / 1e3, n[:100]))
