#!/usr/bin/env bash
# usage: other_trace.sh <config>: serialized kernel trace of one of the other configs: every kernel by total time, with the longest dispatches and their grids
cfg=$1; out=gpurun_out/r5_trace_$cfg; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $cfg 4 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
f=$(find $out/prof -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $out/kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[name[:64]].append((d, r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?')))
tot = sorted(((sum(x[0] for x in v), k) for k, v in agg.items()), reverse=True)
all_us = sum(t for t, _ in tot)
print('all kernels: %.1f ms over 9 iterations (5 warm-up + 4)' % (all_us / 1e3))
for t, k in tot[:32]:
    v = sorted(agg[k], reverse=True)
    print('%-66s n=%5d %5.1f%% avg %6.1f us  top: %s' % (k, len(v), 100 * t / all_us, t / len(v), ', '.join('%.0f us %s/%s' % x for x in v[:3])))
PY
rm -rf $out/prof
cat $out/kernels.txt
