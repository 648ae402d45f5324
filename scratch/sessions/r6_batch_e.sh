#!/usr/bin/env bash
# round 6, GPU session e: grouped weight gradients (GCC_WGRAD_GROUP): kernel + model parity tests, the generators alone, same-box A/B of the step
out=gpurun_out/r6e; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "grouped_weight" -p no:cacheprovider 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_pix2pix_gpu.py tests/test_engine_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -8
for v in 0 1; do echo "== GCC_WGRAD_GROUP=$v"; GCC_WGRAD_GROUP=$v timeout 300 python scratch/unet_ab.py 2>&1 | tail -3; done | tee $out/unet_ab.txt
bash scratch/ab_quick.sh r6e_ab "GCC_WGRAD_GROUP=0" "-" 2>&1 | tee $out/ab.txt
