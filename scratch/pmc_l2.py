"""Per (kernel family, grid) L2 behaviour of the MFMA kernels from one rocprofv3 pass with --kernel-trace --pmc TCC_HIT_sum
TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum:
    python scratch/pmc_l2.py <counter_collection.csv> <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
acc = defaultdict(lambda: defaultdict(float))
seen = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(igemm_kernel<[^>]*>|wgrad_kernel<[^>]*>)', r['Kernel_Name'])
    if not m:
        continue
    key = (m.group(1), r.get('Grid_Size', ''), r.get('LDS_Block_Size', ''))
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen[key]:
        seen[key].add(r['Dispatch_Id'])
        acc[key]['_ns'] += dur.get(r['Dispatch_Id'], 0)
rows = []
for key, c in acc.items():
    n = len(seen[key])
    hit, miss = c.get('TCC_HIT_sum', 0.0), c.get('TCC_MISS_sum', 0.0)
    rows.append((c['_ns'], key, n, c['_ns'] / n / 1e3, hit / max(hit + miss, 1.0), c.get('TCC_EA0_RDREQ_sum', 0.0) / n,
                 c.get('TCP_TCC_READ_REQ_LATENCY_sum', 0.0) / max(c.get('TCP_TCC_READ_REQ_sum', 1.0), 1.0)))
rows.sort(reverse=True)
print('%-40s %-10s %5s %9s %8s %14s %12s' % ('kernel', 'grid', 'n', 'avg us', 'L2 hit', 'EA rdreq/launch', 'rd latency'))
for ns, key, n, us, hr, ea, lat in rows[:24]:
    print('%-40s %-10s %5d %9.1f %8.3f %14.0f %12.0f' % (key[0][:40], key[1], n, us, hr, ea, lat))
