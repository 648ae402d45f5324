#!/usr/bin/env bash
out=gpurun_out/r4bg; mkdir -p $out
timeout 1200 python -m pytest tests/test_replay_gpu.py -q -m gpu -x -k "sagan" -p no:cacheprovider 2>&1 | tail -3
GCC_SAGAN_EARLY_DREAL=1 timeout 1200 python -m pytest tests/test_sagan_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
for cfg in "GCC_SAGAN_EARLY_DREAL=0" "GCC_SAGAN_EARLY_DREAL=1" "GCC_SAGAN_EARLY_DREAL=0" "GCC_SAGAN_EARLY_DREAL=1"; do
  echo "== $cfg"
  env $cfg GCC_BENCH_OTHER=sagan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'), 'streams', v['replay'].get('streams'))"
done
