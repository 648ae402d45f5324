#!/usr/bin/env bash
out=gpurun_out/r4af; mkdir -p $out
timeout 1200 python -m pytest tests/test_srgan_gpu.py tests/test_replay_gpu.py -q -m gpu -x -k "srgan" -p no:cacheprovider 2>&1 | tail -3
for v in 0 1; do
  echo "== GCC_SR_FORK=$v"
  GCC_SR_FORK=$v GCC_BENCH_OTHER=srgan,srgan_96_to_384 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/sr$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())['other_configs']
for k in d: print(k, {a: b for a, b in d[k].items() if a != 'roofline'})"
done
