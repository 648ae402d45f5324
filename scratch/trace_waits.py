"""List the long same-queue waits of the last iteration of a rocprofv3 kernel trace (who waits for whom).
usage: python scratch/trace_waits.py <kernel_trace.csv> [window_ms] [min_wait_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
minw = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']) for r in rows))
t1 = max(e[1] for e in ev)
cut = t1 - win * 1e6
ev = [e for e in ev if e[0] >= cut]
byq = defaultdict(list)
for e in ev:
    byq[e[2]].append(e)
out = []
for q, lst in byq.items():
    for a, b in zip(lst, lst[1:]):
        w = b[0] - a[1]
        if w >= minw * 1e3:
            out.append((a[1], q, w, a[3][:46], b[3][:46]))
for t, q, w, pa, nb in sorted(out):
    # what ran on the other queues during the wait
    busy = defaultdict(int)
    for s, e, qq, n in ev:
        if qq != q:
            o = min(e, t + w) - max(s, t)
            if o > 0:
                busy[qq] += o
    others = ' '.join('q%s:%.0f%%' % (k, 100.0 * v / w) for k, v in sorted(busy.items()))
    print('t=%7.2f ms  queue %s waits %7.1f us  after %-46s before %-46s | meanwhile %s' % ((t - cut) / 1e6, q, w / 1e3, pa, nb, others))
# time per kernel name on the busiest queue
tot = defaultdict(lambda: [0, 0])
bq = max(byq, key=lambda k: sum(e[1] - e[0] for e in byq[k]))
for s, e, q, n in byq[bq]:
    tot[n[:72]][0] += e - s; tot[n[:72]][1] += 1
print('busiest queue %s: %d kernels, busy %.2f ms of %.2f ms' % (bq, len(byq[bq]), sum(v[0] for v in tot.values()) / 1e6, win))
for n, (t, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:22]:
    print('  %7.1f us  %4d x %6.1f us  %s' % (t / 1e3, c, t / c / 1e3, n))
# gap histogram of the busiest queue
lst = sorted(byq[bq])
gaps = [(b[0] - a[1], a[3][:40], b[3][:40]) for a, b in zip(lst, lst[1:])]
for lo, hi in ((0, 3e3), (3e3, 10e3), (10e3, 30e3), (30e3, 150e3), (150e3, 1e12)):
    g = [x for x in gaps if lo <= x[0] < hi]
    print('  gaps %5.0f-%-7.0f us: %4d totalling %7.1f us' % (lo / 1e3, min(hi, 1e9) / 1e3, len(g), sum(x[0] for x in g) / 1e3))
mid = defaultdict(lambda: [0, 0])
for w, pa, nb in gaps:
    if 10e3 <= w < 150e3:
        mid[(pa, nb)][0] += w; mid[(pa, nb)][1] += 1
for (pa, nb), (w, c) in sorted(mid.items(), key=lambda x: -x[1][0])[:14]:
    print('  %7.1f us in %3d gaps  after %-40s before %s' % (w / 1e3, c, pa, nb))
