#!/usr/bin/env bash
# serialized kernel trace of one of the other configs: the last <n> launches, in order (duration, grid, name)
# usage: scratch/other_serial_ctx.sh <outdir under gpurun_out> <config> <n>
out=gpurun_out/$1; w=$2; n=$3; mkdir -p $out; export TMPDIR=/tmp
export GCC_SERIALIZE=1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 2 > $GRAFT_REPO_ROOT/$out/$w.serial.log 2>&1)
t=$(find $out/$w -name '*kernel_trace.csv' | head -1)
python - "$t" "$n" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
for r in rows[-int(sys.argv[2]):]:
    g = int(r['Grid_Size_X'] or 1) // int(r['Workgroup_Size_X'] or 1)
    print('%8.1f us  wgs %6d x%s x%s  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, g, r['Grid_Size_Y'], r['Grid_Size_Z'],
                                          r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]))
PY
rm -rf $out/$w
