#!/usr/bin/env bash
out=gpurun_out/r5_srgan2; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py srgan_96_to_384 4 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_serialized.csv
find $out/prof -name '*kernel_trace.csv' -delete
head -26 $out/kernel_stats_serialized.csv | cut -c1-150
