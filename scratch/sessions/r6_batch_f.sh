#!/usr/bin/env bash
# round 6, GPU session f: grouped weight gradients in the MobileResnet / SRResNet engines: model parity tests, same-box A/B of configs 3 and 5
out=gpurun_out/r6f; mkdir -p $out
timeout 2400 python -m pytest tests/test_cyclegan_gpu.py tests/test_srgan_gpu.py -q -m gpu -x -p no:cacheprovider -k "not 384-16" 2>&1 | tail -4
for m in cyclegan srgan srgan_96_to_384; do bash scratch/ab_other.sh $m "GCC_WGRAD_GROUP=0" "-"; done 2>&1 | tee $out/ab_other2.txt
