#!/usr/bin/env bash
out=gpurun_out/r4ag; mkdir -p $out
for cfg in "GCC_CYCLE_FORK=1" "GCC_CYCLE_FORK=1 GCC_OVERLAP_WGRAD=0" "GCC_CYCLE_FORK=2 GCC_OVERLAP_WGRAD=0" "GCC_CYCLE_FORK=2 GCC_REPLAY_THREADS=8" "GCC_CYCLE_FORK=1 GCC_REPLAY_THREADS=6" "GCC_CYCLE_FORK=0 GCC_OVERLAP_WGRAD=0"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=cyclegan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())['other_configs']['cyclegan']
print({k: v for k, v in d.items() if k != 'roofline'})"
done
