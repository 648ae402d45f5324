#!/usr/bin/env bash
# same-box A/B of two builds of the library, production schedule only: scratch/ab_lib_quick.sh <tag> <baseline.so>
tag=$1; base=$2
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2 3; do
  for v in base new; do
    [ $v = base ] && export GCC_HIP_LIB=$PWD/$base || unset GCC_HIP_LIB
    timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rep $rep %-4s %7.1f img/s %6.3f ms  %d launches' % ('$v', d['value'], d['ms_per_step'], d['launches_per_step']))
" | tee -a $out/ab.txt
  done
done
