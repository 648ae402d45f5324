#!/usr/bin/env bash
out=gpurun_out/r4k; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "tap_stationary" -p no:cacheprovider 2>&1 | tail -15
timeout 600 python scratch/bench_wgrad_ts.py 2>&1 | tee $out/wgrad_ts.txt
bash scratch/ab_quick.sh r4k "-" "GCC_WGRAD_TS=0"
