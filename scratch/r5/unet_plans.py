"""student / teacher U-Net forward + backward alone under candidate tile plans pinned for every convolution call (ops._plan_pinned,
the mechanism behind the GCC_IGEMM_* / GCC_HALO_HC environment pins): us per pass, median of reps.
python3 scratch/r5/unet_plans.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from gcc_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
model, opt = bench.build(16)
train, val = bench.synthetic(16, 0, model.device)
model.set_stream_schedule(False, 'production')
bench.one_step(model, train, val)
torch.cuda.synchronize()
PLANS = [('production', dict(wgrad_wgs_big=128, wgrad_wgs=256)),
         ('alone', dict(pair=1, halo_hc=1)),
         ('pair big_min32', dict(pair=1, big_min=32, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('big_min32', dict(big_min=32, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('big_min32 nk8', dict(big_min=32, big_nk=8, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('pair big_min32 nk8', dict(pair=1, big_min=32, big_nk=8, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('tiles128', dict(tile_families=1, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('tiles256x128 min32', dict(tile_families=2, big_min=32, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('big_min200', dict(big_min=200, wgrad_wgs_big=128, wgrad_wgs=256)),
         ('production wgs256/512', dict()),
         ('production wgs512/1024', dict(wgrad_wgs_big=512, wgrad_wgs=1024))]
for pname, plan in PLANS:
    ops._plan_pinned = dict(plan)
    ops.set_plan()
    for who, m in (('student', model), ('teacher', model.teacher_model)):
        G = m.G
        m.set_input(train)
        f, b = [], []
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
            f.append(e0.elapsed_time(e1) * 1e3); b.append(e1.elapsed_time(e2) * 1e3)
        f.sort(); b.sort()
        print('%-24s %s U-Net: forward %7.1f us, backward %7.1f us' % (pname, who, f[len(f) // 2], b[len(b) // 2]), flush=True)
