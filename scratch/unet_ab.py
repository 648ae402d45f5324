"""student / teacher U-Net forward + backward alone under environment switches set by the caller: prints us per pass
(median of reps).  python3 scratch/unet_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from gcc_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
model.set_stream_schedule(False, 'production')
bench.one_step(model, train, val)
torch.cuda.synchronize()
for who, m in (('student', model), ('teacher', model.teacher_model)):
    G = m.G
    m.set_input(train)
    f, b, n = [], [], []
    for r in range(reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        ops.lib().gcc_launch_count(1)
        e0.record()
        m.forward()
        nf = int(ops.lib().gcc_launch_count(1))
        ctx = m._gctx
        ctx.g_out.fill_(0.01)
        e1.record()
        G.backward(ctx)
        e2.record()
        nb = int(ops.lib().gcc_launch_count(1))
        torch.cuda.synchronize()
        f.append(e0.elapsed_time(e1) * 1e3); b.append(e1.elapsed_time(e2) * 1e3); n.append((nf, nb))
    f.sort(); b.sort()
    print('%s U-Net: forward %.1f us (%d launches), backward %.1f us (%d launches)' % (who, f[len(f) // 2], n[-1][0], b[len(b) // 2], n[-1][1]), flush=True)
