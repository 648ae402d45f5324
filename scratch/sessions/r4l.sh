#!/usr/bin/env bash
out=gpurun_out/r4l; mkdir -p $out
GCC_BENCH_OTHER=srgan_96_to_384 GCC_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/srgan384.json 2> $out/srgan384_shapes.txt
grep -n "other_configs srgan" $out/srgan384_shapes.txt | cut -c1-400
GCC_BENCH_OTHER=cyclegan GCC_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/cyclegan.json 2> $out/cyclegan_shapes.txt
grep -n "other_configs cyc" $out/cyclegan_shapes.txt | cut -c1-400
