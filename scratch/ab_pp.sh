#!/usr/bin/env bash
# A/B of the two-group 256x256 igemm loop on one box: parity first, then the per-shape table both ways
out=gpurun_out/${1:-r02c}; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv_fprop_dgrad_wgrad or true_shapes" 2>&1 | tail -5
for pp in 1 0; do
  GCC_IGEMM_PP=$pp GCC_PROFILE_SHAPES=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_pp$pp.json 2> $out/shapes_pp$pp.txt
  python - <<PY
import json
d=json.load(open('$out/bench_pp$pp.json'))
print('PP=$pp', d['value'], 'img/s', d['ms_per_step'], 'ms; igemm', d['roofline']['achieved'], 'TF/s frac', d['roofline']['frac'])
PY
  grep -E "^igemm_kernel +\('(fprop|dgrad)', 16, (31|64|32), " $out/shapes_pp$pp.txt | grep -v _G | cut -c1-150
done
