#!/usr/bin/env bash
out=gpurun_out/r4aj; mkdir -p $out
for q in 4 5 6 8 12; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  env GPU_MAX_HW_QUEUES=$q GCC_BENCH_OTHER=cyclegan,sagan,srgan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('pix2pix', d['value'])
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'), 'streams', v['replay'].get('streams'))"
done
