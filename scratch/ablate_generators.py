"""What the U-Net passes cost the PRODUCTION schedule (bound the prize first).
Timing only: with `ablate_skip` a generator's forward / backward enqueue nothing (stale buffers, wrong results by design).
    python3 scratch/ablate_generators.py [steps]
prints ms/step of the four-stream production schedule with: nothing skipped, the student's passes skipped, the teacher's,
both -- interleaved A/B/C/D rounds in one process so that box and clock state are shared."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
for _ in range(5):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
arms = [('none', False, False), ('student_G', True, False), ('teacher_G', False, True), ('both', True, True)]
res = {a[0]: [] for a in arms}
for rnd in range(3):
    for name, s, t in arms:
        model.G.ablate_skip, model.teacher_model.G.ablate_skip = s, t
        for _ in range(3):
            bench.one_step(model, train, val)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.one_step(model, train, val)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
model.G.ablate_skip = model.teacher_model.G.ablate_skip = False
base = min(res['none'])
for name, _, _ in arms:
    v = res[name]
    print('skip %-10s ms/step %s  best %.3f  (%.3f ms = %.1f %% of the step)' % (
        name, ' '.join('%.3f' % x for x in v), min(v), base - min(v), 100 * (base - min(v)) / base), flush=True)
