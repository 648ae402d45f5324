"""Feasibility / timing probe: the headline Pix2Pix GCC step (all four streams) captured in one HIP graph
and replayed, against eager enqueue.  Arithmetic caveat of a replay -- by-value kernel arguments (dropout seed, Adam step
count, learning rate) are frozen at capture -- is irrelevant for timing."""
import faulthandler
import os
import sys
import time
faulthandler.enable()

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

batch = int(os.environ.get('BATCH', '16'))
model, opt = bench.build(batch)
dev = model.device
torch.cuda.set_device(dev)
train, val = bench.synthetic(batch, 0, dev)


def step():
    bench.one_step(model, train, val)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_enq = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print('eager: %.2f ms/step (host enqueue %.2f ms)' % ((time.perf_counter() - t0) / 20 * 1e3, t_enq), flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()                  # per-stream workspaces, side streams and contexts of the capture stream
torch.cuda.synchronize()
# events recorded outside the capture must not be waited on inside it: forget the release event of the last eager step
model._teacher_free = None
model._input_ready = None
for d in (train, val):
    d.pop('ready', None)
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
try:
    with torch.cuda.graph(g, stream=s, capture_error_mode=os.environ.get('CAPTURE_MODE', 'thread_local')):
        step()
        ts = model._teacher_stream()
        if ts:
            torch.cuda.current_stream().wait_stream(ts)
        aux = model._aux_stream()
        if aux:
            torch.cuda.current_stream().wait_stream(aux)
except Exception as e:          # noqa: BLE001
    print('capture failed: %r' % (e,), flush=True)
    raise
torch.cuda.synchronize()
print('capture took %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
t_enq = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print('graph replay: %.2f ms/step (host enqueue %.2f ms)' % ((time.perf_counter() - t0) / 20 * 1e3, t_enq), flush=True)
print('losses after replay:', {k: round(v, 4) for k, v in model.get_current_losses().items()})
