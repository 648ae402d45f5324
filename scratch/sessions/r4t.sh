#!/usr/bin/env bash
out=gpurun_out/r4t; mkdir -p $out
GCC_DISTILL_FORK=3 timeout 900 python -m pytest tests/test_pix2pix_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
bash scratch/ab_quick.sh r4t "GCC_DISTILL_FORK=1" "GCC_DISTILL_FORK=3"
