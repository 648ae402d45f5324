#!/usr/bin/env bash
out=gpurun_out/r4as; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "halo" -p no:cacheprovider 2>&1 | tail -4
for v in 2 3; do
  echo "== GCC_IGEMM_HALO=$v"
  GCC_IGEMM_HALO=$v timeout 600 python scratch/diag_vgg.py 2>&1 | grep -v amdgpu.ids | cut -c1-75
done
for v in 2 3; do
  echo "== GCC_IGEMM_HALO=$v"
  env GCC_IGEMM_HALO=$v GCC_BENCH_OTHER=srgan_96_to_384,srgan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'))"
done
