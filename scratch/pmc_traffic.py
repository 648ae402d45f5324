"""HBM traffic of the implicit-GEMM conv kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
    python scratch/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [git head]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request -> x2; both in KiB."""
import collections
import csv
import json
import re
import sys


def family(name):
    m = re.search(r'(igemm_halo_kernel<[^>]*>|igemm_kernel<[^>]*>)', name)
    return m.group(1) if m else None


def table(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        f = family(r['Kernel_Name'])
        if f and r['Counter_Name'] == counter:
            tot[f] += float(r['Counter_Value'])
            n[f] += 1
    return tot, n


ft, fn = table(sys.argv[1], 'FETCH_SIZE')
wt, wn = table(sys.argv[2], 'WRITE_SIZE')
per = {}
for f in sorted(fn):
    fetch, write = ft[f] / fn[f], wt[f] / max(wn[f], 1)
    per[f] = {'launches_sampled': fn[f], 'hbm_bytes_per_launch_corrected': round((2.0 * fetch + write) * 1024.0)}
nf, nw = sum(fn.values()), sum(wn.values())
f, w = sum(ft.values()) / max(nf, 1), sum(wt.values()) / max(nw, 1)
out = {'kernel': 'igemm_kernel + igemm_halo_kernel (all tile variants)', 'launches_sampled': nf, 'fetch_size_kb_avg_raw': f,
       'write_size_kb_avg': w, 'hbm_bytes_per_launch_corrected': (2.0 * f + w) * 1024.0, 'per_family': per,
       'git_head': sys.argv[4] if len(sys.argv) > 4 else None,
       'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; units KiB',
       'command': 'rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 2 --warmup 1 '
                  '--no-cpu-baseline --no-other-configs --no-roofline --serialize-streams'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)
