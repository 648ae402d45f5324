#!/usr/bin/env bash
out=gpurun_out/r3z_ptrace; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline > $GRAFT_REPO_ROOT/$out/log.txt 2>&1)
f=$(find $out/t -name '*kernel_trace.csv' | head -1)
python scratch/trace_busy.py $f 48 > $out/busy.txt; cat $out/busy.txt
python scratch/trace_fill.py $f 32 > $out/fill.txt 2>&1; head -30 $out/fill.txt
python scratch/trace_timeline.py $f 18 200 > $out/timeline.txt 2>&1
rm -rf $out/t
