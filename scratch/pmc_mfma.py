"""MFMA-pipe utilisation and LDS counters per kernel family from rocprofv3 PMC passes (one counter_collection.csv per pass).
    python scratch/pmc_mfma.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]
Definitions (MI355X_MICROARCH.md, 'Per-instruction cycle constants' and 'rocprofv3 PMC slots'):
  SQ_VALU_MFMA_BUSY_CYCLES   sum over the chip's 1024 SIMDs of matrix-pipe busy cycles (16 per v_mfma_f32_16x16x32_bf16)
  GRBM_GUI_ACTIVE            busy cycles summed over the 8 XCDs -> elapsed shader cycles of the dispatch = value / 8
  mfma_util                  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8): fraction of the dispatch's cycles the matrix
                               pipes were busy (at the clock the chip actually held, not the 2.4 GHz of the 2.5 PF peak)
  lds_conflict_frac          = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles / all LDS-array cycles)
Per kernel family (template arguments kept for the MFMA kernels), summed over every dispatch of the pass."""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r'(igemm_halo_kernel<\w+>|igemm_kernel<[^>]*>|wgrad_kernel<[^>]*>|thin_fprop_kernel<\d+>|thin_dgrad_k4s2_kernel<\d+>)', name)
    if m:
        return m.group(1)
    m = re.search(r'([A-Za-z_0-9]+)(<[^(]*)?\(', name)
    return m.group(1) if m else name[:60]


acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        f = family(r['Kernel_Name'])
        acc[f][r['Counter_Name']] += float(r['Counter_Value'])
        calls[f][r['Counter_Name']] += 1
rows = []
for f, c in acc.items():
    n = max(calls[f].values())
    d = {'kernel': f, 'dispatches': n}
    d.update({k: v for k, v in sorted(c.items())})
    if c.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        d['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0), 4)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict_frac'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'], 4)
    if c.get('SQ_BUSY_CU_CYCLES') and c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_active_per_busy_cu_cycle'] = round(c['SQ_LDS_IDX_ACTIVE'] / c['SQ_BUSY_CU_CYCLES'], 4)
    rows.append(d)
rows.sort(key=lambda d: -d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0))
json.dump({'definitions': __doc__, 'kernels': rows}, open(sys.argv[1], 'w'), indent=1)
for d in rows[:14]:
    print('%-46s n=%5d  mfma_util %-7s lds_conflict %-7s' % (d['kernel'][:46], d['dispatches'], d.get('mfma_util', '-'), d.get('lds_conflict_frac', '-')))
