#!/usr/bin/env bash
# production-schedule kernel trace of one of the other configs: per-queue busy time, chip fill, kernel totals
# usage: scratch/other_trace.sh <outdir under gpurun_out> <cyclegan|sagan|srgan|srgan_96_to_384> [window_ms]
out=gpurun_out/$1; w=$2; win=${3:-120}; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/$w -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py $w 10 > $GRAFT_REPO_ROOT/$out/$w.log 2>&1)
tail -1 $out/$w.log
t=$(find $out/$w -name '*kernel_trace.csv' | head -1)
python scratch/trace_summary.py $t > $out/${w}_trace_summary.txt 2>&1
python scratch/trace_fill.py $t $win > $out/${w}_trace_fill.txt 2>&1
cp $(find $out/$w -name '*kernel_stats.csv' | head -1) $out/${w}_kernel_stats.csv
rm -rf $out/$w
head -8 $out/${w}_trace_summary.txt; head -8 $out/${w}_trace_fill.txt
