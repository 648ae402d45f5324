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
