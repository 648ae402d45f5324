#!/usr/bin/env bash
out=gpurun_out/r4bh; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_cyclegan_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "inorm or instance or cyclegan or engine" -p no:cacheprovider 2>&1 | tail -3
GCC_BENCH_OTHER=cyclegan timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline 2> $out/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d['other_configs'].items(): print('  ', k, 'eager', v['ms_per_step'], 'replay', v['replay'].get('ms_per_step'))"
python scratch/host_profile.py cyclegan 10 2>&1 | grep -E "inorm_fwd|inorm_bwd|step\)" | head -5
