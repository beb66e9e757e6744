# One steady-state HRNetPN step (BASELINE config 4 arch) under rocprofv3 --kernel-trace: overlap statistics, busy time per
# queue, and the kernels of the PointNet++ branch in launch order with their queue.  usage: hrnetpn_timeline.sh [width]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp -- python $R/bench.py --arch HRNetPN --width ${1:-18} --steps 6 --warmup 4 --no_cpu_baseline --no_check > /dev/null 2>&1
T=$(find /tmp/hp -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $T 2>&1 | head -60
python - $T <<'PY'
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'bank_pass_' in r[2]]
win = rows[marks[-2]:marks[-1]]
t0 = win[0][0]
print('step window %.2f ms, %d launches' % ((rows[marks[-1]][0] - t0) / 1e6, len(win)))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    return n.split('(')[0][:64]
opt_all = [k for k, r in enumerate(win) if 'multi_tensor' in r[2]]
t_opt = win[opt_all[-1]][0] if opt_all else win[-1][0]
agg = defaultdict(lambda: [0, 0, 0])
for s, e, n, q in win:
    k = short(n)
    agg[(q, k)][0] += e - s
    agg[(q, k)][1] += 1
    if s < t_opt:
        agg[(q, k)][2] += e - s
print('the window runs loss(k) -> backward(k) -> optimizer (+%.2f ms) -> forward(k+1) -> loss(k+1)' % ((t_opt - t0) / 1e6))
print('top kernels by total time (queue, name, ms, calls, ms of it before the optimizer = backward):')
for (q, k), (t, c, tb) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print('  q%s %-64s %8.3f ms %5d   bwd %7.3f' % (q, k, t / 1e6, c, tb / 1e6))
perq_b, perq_f = defaultdict(int), defaultdict(int)
for s, e, n, q in win:
    (perq_b if s < t_opt else perq_f)[q] += e - s
print('busy per queue, backward part (ms):', {q: round(t / 1e6, 2) for q, t in sorted(perq_b.items())})
print('busy per queue, forward part (ms):', {q: round(t / 1e6, 2) for q, t in sorted(perq_f.items())})
perq = defaultdict(int)
for s, e, n, q in win:
    perq[q] += e - s
print('busy per queue (ms):', {q: round(t / 1e6, 2) for q, t in perq.items()})
# who is busy when: busy milliseconds per queue in 2.5 ms bins of the step
nb = int((rows[marks[-1]][0] - t0) / 2.5e6) + 1
bins = defaultdict(lambda: [0.0] * nb)
for s, e, n, q in win:
    a, b_ = s - t0, e - t0
    i = int(a / 2.5e6)
    while a < b_ and i < nb:
        hi = min(b_, (i + 1) * 2.5e6)
        bins[q][i] += (hi - a) / 1e6
        a = hi
        i += 1
for q in sorted(bins):
    print('q%s busy per 2.5 ms bin: %s' % (q, ' '.join('%3.1f' % v for v in bins[q])))
# the start of the next forward: the launches after the optimizer's last multi-tensor kernel, all queues, in start order
opt = [k for k, r in enumerate(win) if 'multi_tensor' in r[2]]
if opt:
    k0 = opt[-1]
    print('after the optimizer (+%.2f ms): queue, start offset us, duration us, kernel' % ((win[k0][0] - t0) / 1e6))
    for s, e, n, q in win[k0:k0 + 70]:
        print('   q%s +%9.1f %8.1f  %s' % (q, (s - win[k0][0]) / 1e3, (e - s) / 1e3, n.split('(')[0][-70:]))
# span of the PointNet++ branch: first fps kernel -> last three_interp / scatter kernel of the step
pn = [(s, e, n, q) for s, e, n, q in win if any(k in n for k in ('fps_kernel', 'ball_query', 'three_nn', 'three_interp', 'gather_rows', 'scatter', 'rowmax'))]
if pn:
    print('PointNet++ op kernels: %d launches, %.3f ms summed, first at +%.2f ms, last ends at +%.2f ms' % (len(pn), sum(e - s for s, e, _, _ in pn) / 1e6, (pn[0][0] - t0) / 1e6, (max(e for _, e, _, _ in pn) - t0) / 1e6))
PY
