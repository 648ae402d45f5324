#!/usr/bin/env bash
out=gpurun_out/r4aw; mkdir -p $out
timeout 1500 python -m pytest tests/test_cyclegan_gpu.py tests/test_replay_gpu.py tests/test_engine_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
GCC_BENCH_OTHER=cyclegan,sagan,srgan timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('pix2pix', d['value'])
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'))"
python scratch/probe_host_unblocked.py 2>&1 | tail -3 | cut -c1-200
