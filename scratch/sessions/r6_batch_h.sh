#!/usr/bin/env bash
# round 6, GPU session h: grouped channel sums: kernel test, model tests of the families that use the collector, same-box A/B
out=gpurun_out/r6h; mkdir -p $out
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_cyclegan_gpu.py tests/test_srgan_gpu.py tests/test_pix2pix_gpu.py -q -m gpu -x -p no:cacheprovider -k "not 384-16" 2>&1 | tail -4
for m in cyclegan srgan; do bash scratch/ab_other.sh $m "GCC_WGRAD_GROUP=0" "-"; done 2>&1 | tee $out/ab_other.txt
