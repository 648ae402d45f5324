"""Which streams share a hardware queue?  With GPU_MAX_HW_QUEUES = 4 (the ROCm default, and the fastest setting for this step by far:
1 queue 695, 2: 775, 3: 805, 4: 905, 5+: 608 images/s) the step's five streams -- student main (Sm), teacher main (Tm), their
weight-gradient side streams (Sw, Tw), the student's auxiliary stream (Sa) -- map onto four queues in order of first use.  This
probe touches them in a given order before the first step and times the step.  usage: python scratch/probe_stream_order.py Sm,Tm,Tw,Sw,Sa"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gcc_amd import ops  # noqa: E402

order = sys.argv[1].split(',') if len(sys.argv) > 1 else []
model, opt = bench.build(16)
dev = model.device
torch.cuda.set_device(dev)
T = model.teacher_model
streams = {'Sm': torch.cuda.current_stream()}
streams['Tm'] = model._teacher_stream()
streams['Sa'] = model._aux_stream()
streams['Sw'] = ops.SideStream.get(dev).stream
with torch.cuda.stream(streams['Tm']):
    streams['Tw'] = ops.SideStream.get(dev).stream
with torch.cuda.stream(streams['Sa']):
    streams['Saw'] = ops.SideStream.get(dev).stream
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
for name in order:                      # first use decides the hardware queue
    with torch.cuda.stream(streams[name]):
        ops.fill(x, 1.0)
    torch.cuda.synchronize()
train, val = bench.synthetic(16, 0, dev)
for _ in range(6):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('%-28s %.2f ms/step  %.1f images/s' % (','.join(order) or '(default)', dt * 1e3, 16 / dt), flush=True)
