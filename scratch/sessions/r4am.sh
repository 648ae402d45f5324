#!/usr/bin/env bash
out=gpurun_out/r4am; mkdir -p $out
for cfg in "GCC_SR_TEACHER_CHAIN_WGRAD=0" "GCC_SR_TEACHER_CHAIN_WGRAD=1" "GCC_SR_TEACHER_CHAIN_WGRAD=0" "GCC_SR_TEACHER_CHAIN_WGRAD=1"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=srgan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'), 'streams', v['replay'].get('streams'))"
done
