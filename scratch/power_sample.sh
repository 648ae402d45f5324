#!/usr/bin/env bash
# socket power and clocks sampled while the bench loop runs (production schedule), then while a serialized loop runs
for mode in "" "--serialize-streams"; do
  python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-roofline $mode > /tmp/b.json 2>/dev/null &
  pid=$!
  sleep 16
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $pid
  python -c "import json;d=json.load(open('/tmp/b.json'));print('mode [$mode]: %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
