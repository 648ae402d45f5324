"""student / teacher U-Net forward + backward alone (for rocprofv3 --kernel-trace): python3 scratch/unet_only.py [student|teacher] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from gcc_amd import ops

who = sys.argv[1] if len(sys.argv) > 1 else 'student'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
model.set_stream_schedule(False, 'production')
m = model if who == 'student' else model.teacher_model
bench.one_step(model, train, val)
torch.cuda.synchronize()
G = m.G
m.set_input(train)
for r in range(reps):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    m.forward()
    ctx = m._gctx
    ctx.g_out.fill_(0.01)
    e1.record()
    G.backward(ctx)
    e2.record()
    torch.cuda.synchronize()
    print('%s U-Net: forward %.1f us, backward %.1f us' % (who, e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3), flush=True)
