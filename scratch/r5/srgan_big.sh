#!/usr/bin/env bash
# SRGAN 96 -> 384 (BASELINE config 5, batch 16): step time under tile-plan pins, and the serialized per-kernel stats of one variant
out=gpurun_out/r5_srgan; mkdir -p $out
export TMPDIR=/tmp
for cfg in "-" "GCC_IGEMM_TILES=1" "GCC_IGEMM_TILES=2" "GCC_IGEMM_BIG_MIN=257" "GCC_HALO_HC=128" "GCC_HALO_HC=1"; do
  [ "$cfg" = "-" ] && envs="" || envs="$cfg"
  env $envs python scratch/other_one.py srgan_96_to_384 12 2>&1 | grep "ms per iteration" | sed "s/^/[$cfg] /"
done
(cd /tmp && GCC_SERIALIZE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/scratch/other_one.py srgan_96_to_384 4 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
find $out/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats_serialized.csv
find $out/prof -name '*kernel_trace.csv' -delete
head -22 $out/kernel_stats_serialized.csv | cut -c1-170
