#!/usr/bin/env bash
out=gpurun_out/r4aq; mkdir -p $out
for cfg in "GCC_IGEMM_BIG_NK=24" "GCC_IGEMM_BIG_NK=16" "GCC_IGEMM_BIG_NK=8" "GCC_IGEMM_BIG_NK=24" "GCC_IGEMM_BIG_NK=16"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=srgan_96_to_384 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('  pix2pix', d['value'])
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'))"
done
