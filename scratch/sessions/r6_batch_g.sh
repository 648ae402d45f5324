#!/usr/bin/env bash
# round 6, GPU session g: size rule of the grouped weight gradients (GCC_WGRAD_GROUP_MAX_UNITS): generators alone, SRGAN 96 -> 384, the headline step
out=gpurun_out/r6g; mkdir -p $out
for v in 8192 20000 1000000; do echo "== GCC_WGRAD_GROUP_MAX_UNITS=$v"; GCC_WGRAD_GROUP_MAX_UNITS=$v timeout 300 python scratch/unet_ab.py 2>&1 | tail -2; done | tee $out/unet_ab.txt
bash scratch/ab_other.sh srgan_96_to_384 "GCC_WGRAD_GROUP=0" "-" 2>&1 | tee $out/ab_srgan384.txt
bash scratch/ab_quick.sh r6g_ab "GCC_WGRAD_GROUP=0" "-" "GCC_WGRAD_GROUP_MAX_UNITS=20000" 2>&1 | tee $out/ab.txt
