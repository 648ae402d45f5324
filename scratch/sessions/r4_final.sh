#!/usr/bin/env bash
# round 4, evidence run on the final tree: tests, bench line, serialized + production rocprof, shapes, PMC, U-Net chains, ablation, batch scan
tag=${1:-r4final}
out=gpurun_out/$tag; mkdir -p $out
bash scratch/gpu_round.sh $tag tests bench prof profdefault shapes pmc
bash scratch/unet_chain.sh $out student
bash scratch/unet_chain.sh $out teacher
timeout 600 python scratch/ablate_generators.py 30 > $out/ablate_generators.txt 2>&1; tail -5 $out/ablate_generators.txt
rm -f $out/batch_scan.txt
for b in 1 2 4 8 16 32; do
  timeout 300 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch %d: %.3f ms/step, %.1f images/s, %d launches' % ($b, d['ms_per_step'], d['value'], d['launches_per_step']))" >> $out/batch_scan.txt
done
cat $out/batch_scan.txt
timeout 300 python scratch/timeline_events.py 2>&1 | tail -52 > $out/phase_timeline_events.txt
timeout 600 python scratch/bench_wgrad_ts.py > $out/wgrad_ts.txt 2>&1
