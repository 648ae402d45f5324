#!/usr/bin/env bash
out=gpurun_out/r4ab; mkdir -p $out
timeout 900 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "two_sides" -p no:cacheprovider 2>&1 | tail -5
GCC_CYCLE_FORK=1 timeout 1200 python -m pytest tests/test_cyclegan_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
for v in 0 1; do
  echo "== GCC_CYCLE_FORK=$v"
  GCC_CYCLE_FORK=$v GCC_BENCH_OTHER=cyclegan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/cyc$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())['other_configs']['cyclegan']
print({k: v for k, v in d.items() if k != 'roofline'})"
done
