"""Is the host ahead of the GPU?  Host wall time per step (enqueue only) against the step time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
model, opt = bench.build(16)
dev = model.device
torch.cuda.set_device(dev)
train, val = bench.synthetic(16, 0, dev)
for _ in range(6):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
ts = model._teacher_stream()
n = 12
drained, tsd, host = [], [], []
t_all = time.perf_counter()
for i in range(n):
    h0 = time.perf_counter()
    bench.one_step(model, train, val)
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t_all) / n * 1e3
print('step %.2f ms; host enqueue per step: %s' % (dt, ' '.join('%.1f' % h for h in host)))
