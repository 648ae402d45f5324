#!/usr/bin/env bash
timeout 900 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "forks" -p no:cacheprovider 2>&1 | tail -3
GCC_ARCH_FREE_EARLY=1 timeout 900 python -m pytest tests/test_pix2pix_gpu.py tests/test_dp_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
bash scratch/ab_quick.sh r4ay "-" "GCC_ARCH_FREE_EARLY=1"
