#!/usr/bin/env bash
# round 6, GPU session b: phase timeline of the default schedule, same-box A/B of GCC_ARCH_EARLY, current rocprof summaries of configs 3-5
out=gpurun_out/r6b; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python scratch/timeline_events.py > $out/timeline_default.txt 2>&1; tail -60 $out/timeline_default.txt
bash scratch/ab_env.sh r6b_ab "-" "GCC_ARCH_EARLY=1" 2>&1 | tee $out/ab_arch_early.txt
bash scratch/other_prof.sh r6b_other 2>&1 | tee $out/other_prof.txt
for w in cyclegan sagan srgan; do f=$(find gpurun_out/r6b_other/$w -name '*kernel_stats.csv' | head -1); cp $f $out/${w}_kernel_stats.csv; done
