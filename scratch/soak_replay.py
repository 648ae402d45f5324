"""Soak run of the replayed iteration: python scratch/soak_replay.py <cyclegan|sagan|srgan> [iterations] -- losses stay finite, memory
does not grow, the iteration time is stable, an invalidate() + re-record now and then (as an epoch boundary does)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.replay import IterationReplay
from gcc_amd.train import SyntheticPairs, attach_teacher
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
batch, argv = bench.OTHER_ARGV[which]
opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
opt.isTrain = True
if getattr(opt, 'teacher_ndf', None) is None:
    opt.teacher_ndf = opt.ndf
cls = get_model_class(opt)
model = cls(opt)
attach_teacher(model, opt, cls)
model.model_train()
data = [{k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in d.items()} for d in SyntheticPairs(opt, 8, 7)]
rp = IterationReplay(model, opt, warmup=3, threads=4, enabled=True)
mem, modes = [], {}
t0 = time.time()
for i in range(n):
    m = rp.step(data[i % 8], data[(i + 3) % 8])
    modes[m] = modes.get(m, 0) + 1
    if i % 250 == 249:
        model.update_learning_rate(1 + i // 250)
        rp.invalidate()
    if i % 100 == 99:
        torch.cuda.synchronize()
        losses = model.get_current_losses()
        ok = all(v == v and abs(v) < 1e6 for v in losses.values())
        mem.append(torch.cuda.memory_allocated() >> 20)
        print('iter %5d  %.2f ms/iter  alloc %d MiB reserved %d MiB  finite %s  %s' % (i + 1, (time.time() - t0) / (i + 1) * 1e3, mem[-1],
              torch.cuda.memory_reserved() >> 20, ok, ' '.join('%s %.3g' % kv for kv in list(losses.items())[:6])), flush=True)
        assert ok, losses
assert max(mem[2:]) - min(mem[2:]) <= 256, mem
print('soak ok: %s, %d iterations %s, allocated %d..%d MiB' % (which, n, modes, min(mem), max(mem)))
