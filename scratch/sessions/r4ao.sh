#!/usr/bin/env bash
out=gpurun_out/r4ao; mkdir -p $out
GCC_TAIL_HALO_HC=1 timeout 900 python -m pytest tests/test_pix2pix_gpu.py -q -m gpu -x -p no:cacheprovider -k "full_config or golden" 2>&1 | tail -3
bash scratch/ab_quick.sh r4ao "-" "GCC_TAIL_HALO_HC=1" "GCC_WGRAD_WGS_BIG=96 GCC_WGRAD_WGS=192" "GCC_WGRAD_WGS_BIG=192 GCC_WGRAD_WGS=384"
