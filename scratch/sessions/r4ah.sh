#!/usr/bin/env bash
out=gpurun_out/r4ah; mkdir -p $out
timeout 1500 python -m pytest tests/test_cyclegan_gpu.py tests/test_replay_gpu.py tests/test_dp_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
GCC_BENCH_OTHER=cyclegan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())['other_configs']['cyclegan']
print({k: v for k, v in d.items() if k != 'roofline'})"
