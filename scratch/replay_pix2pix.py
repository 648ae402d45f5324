"""python scratch/replay_pix2pix.py [steps]: the bench's Pix2Pix iteration eager against replayed (feasibility: dropout seeds are
by-value launch arguments, so a replayed iteration repeats the recorded masks -- timing only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gcc_amd.replay import IterationReplay
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for label, enabled, threads in (('eager', False, 1), ('replay x1', True, 1), ('replay x4', True, 4), ('eager', False, 1), ('replay x4', True, 4)):
    model, opt = bench.build(16)
    train, val = bench.synthetic(16, 0, model.device)
    rp = IterationReplay(model, opt, warmup=4, threads=threads, enabled=enabled)
    for i in range(10):
        rp.step(train, val)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rp.step(train, val)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('pix2pix %-10s %7.3f ms per iteration = %7.1f images/s (host enqueue %6.3f ms)  %s' % (label, ms, 16 / ms * 1e3, th / steps * 1e3, rp.info()), flush=True)
    rp.invalidate()
    del model, rp
    torch.cuda.empty_cache()
