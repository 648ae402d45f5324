#!/usr/bin/env bash
# serialized kernel trace of one of the other configs: every launch of the kernels matching <pattern> in the last iteration
# usage: scratch/other_serial_list.sh <outdir under gpurun_out> <config> <pattern>
out=gpurun_out/$1; w=$2; pat=$3; mkdir -p $out; export TMPDIR=/tmp
export GCC_SERIALIZE=1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 3 > $GRAFT_REPO_ROOT/$out/$w.serial.log 2>&1)
t=$(find $out/$w -name '*kernel_trace.csv' | head -1)
python - "$t" "$pat" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
sel = [r for r in rows if pat in r['Kernel_Name']]
n = len(sel) // 8          # 5 warm-up + 3 iterations
for r in sel[-n:]:
    wg = [int(r['Workgroup_Size_X'] or 1), int(r['Workgroup_Size_Y'] or 1), int(r['Workgroup_Size_Z'] or 1)]
    gr = [int(r['Grid_Size_X'] or 1), int(r['Grid_Size_Y'] or 1), int(r['Grid_Size_Z'] or 1)]
    print('%8.1f us  grid %s wg %s  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, [g // w_ for g, w_ in zip(gr, wg)], wg, r['Kernel_Name'][:90]))
PY
rm -rf $out/$w
