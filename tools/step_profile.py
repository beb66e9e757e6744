"""Post-process a rocprofv3 --kernel-trace CSV on the GPU box: isolate ONE steady-state training
step (the window between the last two launches of the marker kernel) and write a per-kernel
summary (calls, total/avg duration) plus GPU-busy vs wall for that window.

usage: python tools/step_profile.py <kernel_trace.csv> <out_summary.csv> [marker-substring]"""
import csv
import sys
from collections import defaultdict

trace, out = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else 'bank_pass_'
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
assert len(marks) >= 2, 'marker kernel not found twice'
lo, hi = marks[-2], marks[-1]
win = rows[lo:hi]
wall = rows[hi][0] - rows[lo][0]
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in win:
    agg[n][0] += 1
    agg[n][1] += e - s
    busy += e - s
# union of the kernel intervals: with several HIP streams kernels overlap, so "some kernel is running"
# (the GPU-bound floor of the step) is less than the sum of durations
active, cur_s, cur_e = 0, None, None
for s, e, _ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            active += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    active += cur_e - cur_s
items = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(out, 'w') as f:
    w = csv.writer(f)
    w.writerow(['# one steady-state step: window between the last two launches of', marker])
    w.writerow(['# wall_ms', round(wall / 1e6, 3), 'gpu_busy_ms', round(busy / 1e6, 3), 'kernel_launches', len(win),
                'gpu_active_ms (union over streams)', round(active / 1e6, 3)])
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'PercentOfBusy'])
    for n, (c, t) in items:
        w.writerow([n[:160], c, round(t / 1e3, 2), round(t / 1e3 / c, 2), round(100.0 * t / busy, 2)])
print('step wall %.2f ms, gpu busy (sum) %.2f ms, gpu active (union) %.2f ms, %d launches' % (wall / 1e6, busy / 1e6, active / 1e6, len(win)))
for n, (c, t) in items[:14]:
    print('%8.2f ms %6d  %s' % (t / 1e6, c, n[:110]))
