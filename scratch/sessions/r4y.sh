#!/usr/bin/env bash
out=gpurun_out/r4y; mkdir -p $out
timeout 900 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "forks or pix2pix" -p no:cacheprovider 2>&1 | tail -5
GCC_ARCH_FORK=1 timeout 900 python -m pytest tests/test_pix2pix_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
bash scratch/ab_quick.sh r4y "-" "GCC_ARCH_FORK=1"
