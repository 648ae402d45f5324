"""Phase timeline of the production schedule WITHOUT a profiler: HIP events at the phase boundaries of both networks (recorded on
whatever stream the phase runs on), a few iterations, times relative to the start of the student's optimize_parameters.
The profiler's kernel trace slows the host enough to change the overlap; this does not (about 20 events per iteration)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gcc_amd.models import Pix2Pix as P  # noqa: E402

DP = os.environ.get('GCC_TL_DP', '')          # 'torch' / 'native': one rank of an RCCL process group, the model on its world > 1 code paths
if DP:
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29578', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', GCC_DP_FORCE_BUCKETS='1', GCC_DP_COMM=DP)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
model, opt = bench.build(16)
if DP:
    model._world = model.teacher_model._world = 2
dev = model.device
torch.cuda.set_device(dev)
train, val = bench.synthetic(16, 0, dev)
T = model.teacher_model
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    marks.append((name, e))


def wrap_gen(obj, attr, name):
    orig = getattr(obj, attr)

    def g(*a, **k):
        mark(name + ' begin')
        yield from orig(*a, **k)
        mark(name + ' end')
    setattr(obj, attr, g)


def wrap_fn(obj, attr, name):
    orig = getattr(obj, attr)

    def f(*a, **k):
        mark(name + ' begin')
        r = orig(*a, **k)
        mark(name + ' end')
        return r
    setattr(obj, attr, f)


wrap_gen(T, '_iteration_steps', 'T.iteration')
wrap_gen(T, '_pre_join_steps', 'T.pre-join (G fwd, D step, head)')
wrap_fn(T, '_backward_G_tail', 'T.G backward')
wrap_gen(model, '_pre_join_steps', 'S.pre-join (G fwd, D step, head)')
wrap_fn(model, '_backward_G_tail', 'S.tail (T.D on fake, distill, G backward)')
wrap_fn(model, '_apply_G_update', 'S.adam G + repack')
wrap_fn(model, 'optimizer_netD_arch', 'S.arch step')
wrap_fn(model, 'backward_D_arch', 'S.arch backward (incl. its 2 D forwards)')
wrap_fn(model, '_distill_teacher_d_terms', 'S.tail: teacher D over the fake + its terms')
wrap_fn(model, '_distill_generator_terms', 'S.tail: generator-feature terms (aux stream when forked)')
wrap_fn(model.G, 'backward', 'S.G backward')
wrap_fn(T.G, 'backward', 'T.G backward (engine)')
wrap_fn(model, 'forward', 'S.forward')
wrap_fn(T, 'forward', 'T.forward')
wrap_gen(model, '_backward_D_steps', 'S.D step')
wrap_gen(T, '_backward_D_steps', 'T.D step')
wrap_gen(model, '_backward_G_head_steps', 'S.head')
wrap_gen(T, '_backward_G_head_steps', 'T.head')
wrap_fn(model, 'get_D_arch_diff', 'S.arch D forwards + diff')
if DP:
    wrap_fn(model, 'finish_G_update', 'S.finish_G_update')
    wrap_fn(T, 'finish_G_update', 'T.finish_G_update (deferred Adam G + repack)')
    wrap_fn(model, '_allreduce', 'S._allreduce (finish buckets)')
    wrap_fn(T, '_allreduce', 'T._allreduce (finish buckets)')
orig_free = model._mark_teacher_free


def free_():
    orig_free()
    mark('S: teacher free (main stream)')
model._mark_teacher_free = free_
orig_rel = model._release_teacher_stream


def rel_(ts, *a, **k):
    orig_rel(ts, *a, **k)
    with torch.cuda.stream(ts):
        mark('T stream released (after its waits)')
model._release_teacher_stream = rel_
orig_diff = T.get_D_arch_diff


def tdiff(*a, **k):
    r = orig_diff(*a, **k)
    mark('T.arch forward + diff end')
    return r
T.get_D_arch_diff = tdiff

for _ in range(6):
    bench.one_step(model, train, val)
torch.cuda.synchronize()
for it in range(3):
    marks.clear()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    bench.one_step(model, train, val)
    t1 = torch.cuda.Event(enable_timing=True)
    t1.record()
    torch.cuda.synchronize()
    print('--- iteration %d: %.2f ms (main-stream events)' % (it, t0.elapsed_time(t1)))
    for name, e in marks:
        print('  %7.2f ms  %s' % (t0.elapsed_time(e), name))
