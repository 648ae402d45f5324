#!/usr/bin/env bash
# SRGAN 96 -> 384: step time with / without the ring-walk + thin-output routes, then serialized kernel statistics of the current tree
out=gpurun_out/r5_srgan4; mkdir -p $out
export TMPDIR=/tmp
python scratch/other_one.py srgan_96_to_384 12 > $out/step_default.txt 2>&1; tail -1 $out/step_default.txt
GCC_IGEMM_THIN=0 python scratch/other_one.py srgan_96_to_384 12 > $out/step_nothin.txt 2>&1; tail -1 $out/step_nothin.txt
python scratch/other_one.py srgan 20 > $out/step_small.txt 2>&1; tail -1 $out/step_small.txt
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py srgan_96_to_384 4 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_serialized.csv
find $out/prof -name '*kernel_trace.csv' -delete
head -30 $out/kernel_stats_serialized.csv | cut -c1-160
