#!/usr/bin/env bash
out=gpurun_out/r4bc; mkdir -p $out
timeout 1200 python -m pytest tests/test_sagan_gpu.py tests/test_replay_gpu.py -q -m gpu -x -k "sagan or spectral" -p no:cacheprovider 2>&1 | tail -4
for cfg in "GCC_SN_FUSED_PACK=0" "GCC_SN_FUSED_PACK=1" "GCC_SN_FUSED_PACK=0" "GCC_SN_FUSED_PACK=1"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=sagan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'), 'launches', v['launches_per_step'])"
done
