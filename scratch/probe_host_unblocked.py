"""Host enqueue time of one Pix2Pix GCC step with the GPU idle at its start (nothing to wait for) against the steady state:
if the two agree the host is the pace-maker; if the idle-start figure is much smaller, launches block on the device."""
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
idle, drain = [], []
for i in range(8):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    bench.one_step(model, train, val)
    h1 = time.perf_counter()
    torch.cuda.synchronize()
    h2 = time.perf_counter()
    idle.append((h1 - h0) * 1e3); drain.append((h2 - h0) * 1e3)
print('idle start: host enqueue per step %s ; until drained %s' % (' '.join('%.1f' % h for h in idle), ' '.join('%.1f' % h for h in drain)))
host = []
t_all = time.perf_counter()
for i in range(12):
    h0 = time.perf_counter()
    bench.one_step(model, train, val)
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
print('steady: step %.2f ms; host enqueue per step: %s' % ((time.perf_counter() - t_all) / 12 * 1e3, ' '.join('%.1f' % h for h in host)))
# per-phase host stamps of one idle-start step
import gcc_amd.models.Pix2Pix as P
marks = []
orig_tail = model._backward_G_tail
def tail(ts=None):
    marks.append(('tail begin', time.perf_counter())); r = orig_tail(ts); marks.append(('tail end', time.perf_counter())); return r
model._backward_G_tail = tail
orig_arch = model.optimizer_netD_arch
def arch():
    marks.append(('arch begin', time.perf_counter())); r = orig_arch(); marks.append(('arch end', time.perf_counter())); return r
model.optimizer_netD_arch = arch
torch.cuda.synchronize()
h0 = time.perf_counter()
bench.one_step(model, train, val)
h1 = time.perf_counter()
torch.cuda.synchronize()
print('phases (ms from step start, idle start):', ' | '.join('%s %.2f' % (k, (t - h0) * 1e3) for k, t in marks), '| step end %.2f' % ((h1 - h0) * 1e3))
