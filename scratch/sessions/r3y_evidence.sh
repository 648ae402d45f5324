#!/usr/bin/env bash
# the measurements behind profiles/r3y_*: one box, one after the other
out=gpurun_out/r3y_ev; mkdir -p $out
timeout 200 python scratch/inorm_bench.py 2>&1 | grep -v amdgpu > $out/inorm_bench.txt
timeout 200 python scratch/inorm_stamps.py 2>&1 | grep -v amdgpu > $out/inorm_stamps.txt
bash scratch/ab_other.sh cyclegan "GCC_INORM_GRID=0 GCC_INORM_FUSED_MAX_HW=4096" "-" > $out/cyclegan_inorm_ab.txt 2>&1
for m in cyclegan sagan srgan; do timeout 400 python scratch/replay_bench.py $m 30 2>&1 | grep -v amdgpu | tail -4; done > $out/replay_bench.txt
for m in cyclegan sagan srgan; do GCC_REPLAY_TIMING=1 timeout 300 python scratch/replay_bench.py $m 3 replayx1 2>&1 | grep -v amdgpu | tail -2; done > $out/replay_host_timing.txt
cat $out/replay_bench.txt
