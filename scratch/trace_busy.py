"""python scratch/trace_busy.py <kernel_trace.csv> <window_ms>: over the last window of a rocprofv3 kernel trace -- wall time, time
with >= 1 kernel running (union over queues), busy time and launches per queue, the largest idle gaps of the union"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']) for r in rows]
t1 = max(e[1] for e in ev)
ev = sorted(e for e in ev if e[0] >= t1 - win)
t0 = ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
for s, e, q, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n[:60])); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('window %.2f ms, %d kernels, >=1 kernel running %.2f ms (%.0f %%), sum of kernel durations %.2f ms' % (
    (t1 - t0) / 1e6, len(ev), busy / 1e6, 100.0 * busy / (t1 - t0), sum(e - s for s, e, _, _ in ev) / 1e6))
perq = defaultdict(lambda: [0, 0])
for s, e, q, n in ev:
    perq[q][0] += e - s; perq[q][1] += 1
for q, (b, c) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print('  queue %s: %.2f ms busy, %d kernels (avg %.1f us)' % (q, b / 1e6, c, b / c / 1e3))
gaps.sort(reverse=True)
print('  idle gaps of the union: %d, total %.2f ms, top: %s' % (len(gaps), sum(g for g, _ in gaps) / 1e6, ', '.join('%.0f us before %s' % (g / 1e3, n[:30]) for g, n in gaps[:6])))
