"""HBM traffic of the igemm kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
    python scratch/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request -> x2; both in KiB."""
import csv
import json
import sys


def avg(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if 'igemm_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value'])
            n += 1
    return tot / max(n, 1), n


f, nf = avg(sys.argv[1], 'FETCH_SIZE')
w, nw = avg(sys.argv[2], 'WRITE_SIZE')
out = {'kernel': 'igemm_kernel (all tile variants)', 'launches_sampled': nf, 'fetch_size_kb_avg_raw': f, 'write_size_kb_avg': w,
       'hbm_bytes_per_launch_corrected': (2.0 * f + w) * 1024.0,
       'correction': 'gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; units KiB',
       'command': 'rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --steps 2 --warmup 1 '
                  '--no-cpu-baseline --no-roofline'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)
