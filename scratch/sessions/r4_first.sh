#!/usr/bin/env bash
# round 4, first GPU call: baseline bench line of the round-3 tree, generator ablation, batch scan
out=gpurun_out/r4a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py --no-other-configs > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; head -c 600 $out/bench.json; echo
timeout 600 python scratch/ablate_generators.py 30 > $out/ablate_generators.txt 2>&1; cat $out/ablate_generators.txt | tail -6
for b in 1 2 4 8 16 32; do
  timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %d: %.3f ms/step, %.1f images/s, %d launches' % ($b, d['ms_per_step'], d['value'], d['launches_per_step']))" >> $out/batch_scan.txt
done
cat $out/batch_scan.txt
