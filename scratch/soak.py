"""Soak run of the training loop on synthetic pairs: losses stay finite, device memory does not grow, and the iteration time
is stable.  python scratch/soak.py [iterations]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
g = torch.Generator(device=model.device).manual_seed(5)
mem = []
t0 = time.time()
for i in range(n):
    if i % 20 == 0:          # fresh data now and then: the losses must not depend on a memorised batch
        for d in (train, val):
            for k in ('A', 'B'):
                d[k].copy_(torch.rand(d[k].shape, generator=g, device=model.device) * 2 - 1)
    bench.one_step(model, train, val)
    if i % 50 == 49:
        torch.cuda.synchronize()
        losses = model.get_current_losses()
        ok = all(v == v and abs(v) < 1e6 for v in losses.values())
        mem.append(torch.cuda.memory_allocated() >> 20)
        print('iter %4d  %.1f ms/iter  alloc %d MiB  reserved %d MiB  finite %s  %s' % (
            i + 1, (time.time() - t0) / (i + 1) * 1e3, mem[-1], torch.cuda.memory_reserved() >> 20, ok,
            ' '.join('%s %.3g' % kv for kv in losses.items())), flush=True)
        assert ok, losses
model.update_learning_rate(1)
assert max(mem[1:]) - min(mem[1:]) <= 64, mem
print('soak ok: %d iterations, allocated memory %d..%d MiB' % (n, min(mem), max(mem)))
