"""Chip fill over time from a rocprofv3 kernel trace of bench.py (production schedule): every running kernel contributes
fill = min(1, workgroups / 256 CUs) (a launch with fewer workgroups than CUs cannot use the whole chip; one with more is taken
as chip-filling), summed over the concurrently running kernels and clipped at 1.  Prints the time-weighted histogram of the
summed fill over the last iteration(s) and, for the low-fill time, which kernels were running.
usage: python scratch/trace_fill.py <kernel_trace.csv> [window_ms]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 36.0


def wgs(r):
    n = 1
    for a, b in (('Grid_Size_X', 'Workgroup_Size_X'), ('Grid_Size_Y', 'Workgroup_Size_Y'), ('Grid_Size_Z', 'Workgroup_Size_Z')):
        g, w = int(r.get(a, 1) or 1), int(r.get(b, 1) or 1)
        n *= max(1, (g + w - 1) // max(w, 1))
    return n


def short(n):
    import re
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]*>)?)', n)
    return (m.group(1) if m else n)[:56]


ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), min(1.0, wgs(r) / 256.0), short(r['Kernel_Name']), r.get('Queue_Id', '0')))
t1 = max(e[1] for e in ev)
cut = t1 - win * 1e6
ev = [e for e in ev if e[0] >= cut]
pts = []
for i, (s, e, f, n, q) in enumerate(ev):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort()
active = set()
hist = defaultdict(float)
low_by = defaultdict(float)
mid_by = defaultdict(float)
alone_by = defaultdict(float)
prev = pts[0][0]
for t, kind, i in pts:
    dt = t - prev
    if dt > 0:
        fill = min(1.0, sum(ev[j][2] for j in active))
        b = 'idle' if not active else ('<0.25' if fill < 0.25 else '<0.5' if fill < 0.5 else '<1' if fill < 1.0 else 'full')
        hist[b] += dt
        if active and fill < 0.5:
            for j in active:
                low_by[(ev[j][3], ev[j][4])] += dt
        elif active and fill < 1.0:
            for j in active:
                mid_by[(ev[j][3], ev[j][4])] += dt
        if len(active) == 1:
            for j in active:
                alone_by[(ev[j][3], ev[j][4], ev[j][2] >= 1.0)] += dt
    prev = t
    if kind == 1:
        active.add(i)
    else:
        active.discard(i)
tot = sum(hist.values())
print('window %.2f ms' % (tot / 1e6))
for b in ('idle', '<0.25', '<0.5', '<1', 'full'):
    print('  summed fill %-6s %7.2f ms  %5.1f %%' % (b, hist[b] / 1e6, 100.0 * hist[b] / tot))
print('kernels running while the summed fill is below 0.5 (time, queue):')
for (n, q), t in sorted(low_by.items(), key=lambda x: -x[1])[:25]:
    print('  %7.2f ms  q%s  %s' % (t / 1e6, q, n))
print('kernels running while the summed fill is in [0.5, 1) (time, queue):')
for (n, q), t in sorted(mid_by.items(), key=lambda x: -x[1])[:20]:
    print('  %7.2f ms  q%s  %s' % (t / 1e6, q, n))
print('time with exactly ONE kernel running: %.2f ms of which chip-filling launches %.2f ms; the others:' % (
    sum(alone_by.values()) / 1e6, sum(t for (n, q, f), t in alone_by.items() if f) / 1e6))
for (n, q, f), t in sorted(((k, v) for k, v in alone_by.items() if not k[2]), key=lambda x: -x[1])[:20]:
    print('  %7.2f ms  q%s  %s' % (t / 1e6, q, n))

cnt = defaultdict(int)
for s_, e_, f_, n_, q_ in ev:
    cnt[n_] += 1
# exact per-iteration census: everything launched between the last two arch_coeffs launches (one per GCC iteration)
marks = sorted(e[0] for e in ev if e[3].startswith('arch_coeffs'))
if len(marks) >= 2:
    a_, b_ = marks[-2], marks[-1]
    one = [e for e in ev if a_ <= e[0] < b_]
    print('launches per iteration (between the last two arch_coeffs launches, %.2f ms apart): %d' % ((b_ - a_) / 1e6, len(one)))
    cnt = defaultdict(int)
    for s_, e_, f_, n_, q_ in one:
        cnt[n_] += 1
print('launches in the window: %d' % len(ev))
for n_, c_ in sorted(cnt.items(), key=lambda x: -x[1])[:40]:
    print('  %5d  %s' % (c_, n_))
