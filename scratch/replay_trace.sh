#!/usr/bin/env bash
# is the replayed iteration bound by the GPU or by the issuing thread?  kernel trace of replay x1 / x4 -> trace_busy.py
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for m in sagan srgan cyclegan; do
  for mode in replayx1 replayx4; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/${m}_$mode -- python3 $GRAFT_REPO_ROOT/scratch/replay_bench.py $m 20 $mode > $GRAFT_REPO_ROOT/$out/${m}_$mode.log 2>&1)
    grep "ms per iteration" $out/${m}_$mode.log
    f=$(find $out/${m}_$mode -name '*kernel_trace.csv' | head -1)
    python scratch/trace_busy.py $f 100
    rm -rf $out/${m}_$mode
  done
done
