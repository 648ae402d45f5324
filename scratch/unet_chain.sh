#!/usr/bin/env bash
# kernel-by-kernel listing of the student / teacher U-Net forward + backward alone (rocprofv3 --kernel-trace of scratch/unet_only.py)
# usage: scratch/unet_chain.sh <outdir> [student|teacher]
out=$1; who=${2:-student}; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/chain_$who -- python3 $GRAFT_REPO_ROOT/scratch/unet_only.py $who 4 > $GRAFT_REPO_ROOT/$out/unet_only_$who.log 2>&1)
f=$(find $out/chain_$who -name '*kernel_trace.csv' | head -1)
python - "$f" > $out/unet_${who}_chain.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# the last forward + backward pair: from the last-but-one nhwc_copy_group (set_input) on
# the last forward + backward pair: from the last-but-one image-layer forward kernel (the last one is the backward's dgrad of u0)
starts = [i for i, r in enumerate(rows) if 'thin_fprop_kernel' in r['Kernel_Name']]
first = starts[-2] if len(starts) >= 2 else 0
for r in rows[max(0, first - 4):]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%10.1f us  dur %7.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:110]))
PY
rm -rf $out/chain_$who
grep "U-Net" $out/unet_only_$who.log | tail -3
