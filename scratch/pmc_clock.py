"""Effective clock of the long igemm launches from a rocprofv3 PMC pass (GRBM_GUI_ACTIVE) joined with the kernel trace:
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, 'DVFS give-back'; reads high on dispatches < 0.3 ms).
    python scratch/pmc_clock.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import csv
import json
import sys

dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and 'igemm_kernel<256, 256' in r['Kernel_Name'] and r['Dispatch_Id'] in dur:
        ns = dur[r['Dispatch_Id']][0]
        if ns > 150000:
            rows.append((ns, float(r['Counter_Value']) / 8.0 / ns))
rows.sort()
clk = sorted(c for _, c in rows)
out = {'kernel': 'igemm_kernel<256,256> launches longer than 150 us', 'launches': len(rows),
       'effective_clock_ghz_median': clk[len(clk) // 2] if clk else None,
       'effective_clock_ghz_min_max': [clk[0], clk[-1]] if clk else None,
       'avg_duration_us': sum(n for n, _ in rows) / max(len(rows), 1) / 1e3,
       'method': 'GRBM_GUI_ACTIVE / 8 / duration; the quotient reads high on short dispatches (guide: < 0.3 ms)'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(out)
