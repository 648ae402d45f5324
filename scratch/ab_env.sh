#!/usr/bin/env bash
# same-box A/B of tuning options through their GCC_* environment defaults: scratch/ab_env.sh <tag> "ENV1=a ENV2=b" "ENV3=c" ...
# (an argument "-" is the default configuration); every variant runs bench.py --steps 30 twice, order A B C ... A B C
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    [ "$cfg" = "-" ] && envs="" || envs="$cfg"
    env $envs GCC_PROFILE_SHAPES=1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > $out/v${i}_r$rep.json 2> $out/v${i}_r$rep.txt
    python - <<PY
import json
try:
    d=json.load(open('$out/v${i}_r$rep.json'))
    r=d['roofline']
    print('rep $rep [%-40s] %7.1f img/s %6.2f ms | igemm %6.1f TF/s (%.4f) wgrad %6.1f TF/s' % ('$cfg', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['per_kernel'].get('wgrad_kernel (+ slab reduce)',{}).get('tflops',0)))
except Exception as e:
    print('rep $rep [$cfg] failed', e)
PY
  done
done
