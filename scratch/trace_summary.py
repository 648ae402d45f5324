"""Summarise a rocprofv3 kernel trace of bench.py: per-queue busy time, union busy time, idle gaps.
usage: python scratch/trace_summary.py <kernel_trace.csv> [window_ms: the last so many milliseconds of the trace]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '0'), r['Kernel_Name']) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - win * 1e6                 # steady state only
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
busy = defaultdict(int)
for s, e, q, n in ev:
    busy[q] += e - s
# union
u, cur_s, cur_e = 0, None, None
gaps = []
for s, e, q, n in ev:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        u += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
print('span %.2f ms, union busy %.2f ms (%.1f%%), sum of kernels %.2f ms' % (span / 1e6, u / 1e6, 100.0 * u / span, sum(busy.values()) / 1e6))
for q, b in sorted(busy.items(), key=lambda x: -x[1]):
    print('  queue %s: busy %.2f ms (%.1f%%)' % (q, b / 1e6, 100.0 * b / span))
gaps.sort(reverse=True)
print('idle gaps: %d, total %.2f ms; > 5 us: %d totalling %.2f ms' % (len(gaps), sum(g for g, _ in gaps) / 1e6,
      sum(1 for g, _ in gaps if g > 5000), sum(g for g, _ in gaps if g > 5000) / 1e6))
by = defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n[:60]][0] += g; by[n[:60]][1] += 1
for n, (g, c) in sorted(by.items(), key=lambda x: -x[1][0])[:12]:
    print('  %7.1f us over %4d gaps before %s' % (g / 1e3, c, n))
cnt = defaultdict(lambda: [0, 0])
for s_, e_, q_, n_ in ev:
    k = n_[:70]
    cnt[k][0] += 1; cnt[k][1] += e_ - s_
print('launches in the window: %d' % len(ev))
for k, (c, t) in sorted(cnt.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 0]:
    print('  %5d x %7.1f us  %s' % (c, t / c / 1e3, k))
# per queue: gaps between consecutive kernels of the same queue (dispatch latency when short, dependency waits when long)
byq = defaultdict(list)
for s_, e_, q_, n_ in ev:
    byq[q_].append((s_, e_, n_))
for q_, lst in sorted(byq.items()):
    lst.sort()
    short = [b[0] - a[1] for a, b in zip(lst, lst[1:]) if 0 <= b[0] - a[1] < 30000]
    long_ = [b[0] - a[1] for a, b in zip(lst, lst[1:]) if b[0] - a[1] >= 30000]
    print('  queue %s: %d kernels; %d back-to-back gaps (< 30 us) totalling %.2f ms (avg %.1f us); %d longer waits totalling %.2f ms'
          % (q_, len(lst), len(short), sum(short) / 1e6, (sum(short) / max(len(short), 1)) / 1e3, len(long_), sum(long_) / 1e6))
