#!/usr/bin/env bash
timeout 900 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "forks" -p no:cacheprovider 2>&1 | tail -3
GCC_FWD_SIDE=1 timeout 900 python -m pytest tests/test_pix2pix_gpu.py tests/test_replay_gpu.py -q -m gpu -x -k "pix2pix or golden or oracle" -p no:cacheprovider 2>&1 | tail -3
bash scratch/ab_quick.sh r4az "-" "GCC_FWD_SIDE=1"
