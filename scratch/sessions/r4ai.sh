#!/usr/bin/env bash
out=gpurun_out/r4ai; mkdir -p $out
for cfg in "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 GCC_CYCLE_CHAIN_WGRAD=0 GCC_REPLAY_THREADS=8" "GPU_MAX_HW_QUEUES=6 GCC_CYCLE_CHAIN_WGRAD=0 GCC_CYCLE_FORK=1 GCC_REPLAY_THREADS=6"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=cyclegan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('pix2pix', d['value'], {k: v for k, v in d['other_configs']['cyclegan'].items() if k != 'roofline'})"
done
