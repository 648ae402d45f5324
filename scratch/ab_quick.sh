#!/usr/bin/env bash
# same-box A/B through environment switches, production schedule only (no roofline step): scratch/ab_quick.sh <tag> "ENV=.." "-" ...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    [ "$cfg" = "-" ] && envs="" || envs="$cfg"
    env $envs timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline 2> $out/v${i}_r$rep.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('rep $rep [%-44s] %7.1f img/s %6.3f ms  %d launches' % ('$cfg', d['value'], d['ms_per_step'], d['launches_per_step']))
except Exception as e:
    print('rep $rep [$cfg] failed', e)
" | tee -a $out/ab.txt
  done
done
