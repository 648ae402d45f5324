#!/usr/bin/env bash
# every launch on one stream (a kernel's duration is its own): rocprofv3 kernel stats of one of the other configs
# usage: scratch/other_serial.sh <outdir under gpurun_out> <cyclegan|sagan|srgan|srgan_96_to_384>
out=gpurun_out/$1; w=$2; mkdir -p $out; export TMPDIR=/tmp
export GCC_SERIALIZE=1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 10 > $GRAFT_REPO_ROOT/$out/$w.serial.log 2>&1)
tail -1 $out/$w.serial.log
cp $(find $out/$w -name '*kernel_stats.csv' | head -1) $out/${w}_kernel_stats_serialized.csv
rm -rf $out/$w
