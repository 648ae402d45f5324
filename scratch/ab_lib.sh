#!/usr/bin/env bash
# same-box A/B of two builds of the library: scratch/ab_lib.sh <tag> <baseline.so> ; the working tree's library is "new".
# Prints images/s, step time and the generator pass times (bracketed step) of each run, order base new base new.
tag=$1; base=$2
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2; do
  for v in base new; do
    [ $v = base ] && export GCC_HIP_LIB=$PWD/$base || unset GCC_HIP_LIB
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > $out/${v}_r$rep.json
    python - <<PY
import json
d=json.load(open('$out/${v}_r$rep.json')); r=d['roofline']; g=r['generator']
print('rep $rep %-4s %7.1f img/s %6.2f ms | igemm %6.1f TF/s (%.4f) | student G pass %.3f ms conv %.3f | teacher G pass %.3f conv %.3f' % (
    '$v', d['value'], d['ms_per_step'], r['achieved'], r['frac'], g['student_G']['pass_ms'], g['student_G']['conv_ms'], g['teacher_G']['pass_ms'], g['teacher_G']['conv_ms']))
PY
  done
done
