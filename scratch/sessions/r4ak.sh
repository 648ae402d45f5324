#!/usr/bin/env bash
out=gpurun_out/r4ak; mkdir -p $out
for cfg in "GCC_OVERLAP_WGRAD=1" "GCC_OVERLAP_WGRAD=0"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=sagan,srgan,srgan_96_to_384 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'), 'streams', v['replay'].get('streams'))"
done
