"""Feasibility probe: capture CycleGAN's student forward + backward_G (about 2000 launches) in a HIP graph and compare the
replay time with eager enqueue."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs

argv = ['--dataroot', 'synthetic', '--model', 'cyclegan', '--ngf', '64', '--ndf', '64', '--gpu_ids', '0', '--batch_size', '1']
opt = options.parse(argv)
opt.isTrain = True
model = get_model_class(opt)(opt)
model.model_train()
data = list(SyntheticPairs(opt, 2, 7))
model.set_input(data[0])


def work():
    model.forward()
    model.optimizer_G.zero_grad()
    model.backward_G()


for _ in range(3):
    work()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    work()
t_enq = (time.time() - t0) / 10 * 1e3
torch.cuda.synchronize()
print('eager: %.2f ms (host enqueue %.2f ms)' % ((time.time() - t0) / 10 * 1e3, t_enq), flush=True)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    work()                      # warm the per-stream workspaces / side streams of the capture stream
torch.cuda.synchronize()
t0 = time.time()
with torch.cuda.graph(g, stream=s):
    work()
torch.cuda.synchronize()
print('capture took %.1f ms' % ((time.time() - t0) * 1e3), flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    g.replay()
t_enq = (time.time() - t0) / 10 * 1e3
torch.cuda.synchronize()
print('graph replay: %.2f ms (host enqueue %.2f ms)' % ((time.time() - t0) / 10 * 1e3, t_enq), flush=True)
