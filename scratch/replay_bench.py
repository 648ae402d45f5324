"""python scratch/replay_bench.py <cyclegan|sagan|srgan> [steps]: bench.py's other configs, eager against replayed (1 and 4 host
threads), in one process each"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gcc_amd import ops
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.replay import IterationReplay
from gcc_amd.train import SyntheticPairs, attach_teacher
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
batch, argv = bench.OTHER_ARGV[which]
modes = (('eager', False, 1), ('replay x1', True, 1), ('replay x2', True, 2), ('replay x4', True, 4))
if len(sys.argv) > 3:
    modes = [m for m in modes if m[0].replace(' ', '') in sys.argv[3].split(',')]
for label, enabled, threads in modes:
    opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
    opt.isTrain = True
    if getattr(opt, 'teacher_ndf', None) is None:
        opt.teacher_ndf = opt.ndf
    cls = get_model_class(opt)
    model = cls(opt)
    attach_teacher(model, opt, cls)
    model.model_train()
    data = list(SyntheticPairs(opt, 4, 7))
    rp = IterationReplay(model, opt, warmup=3, threads=threads, enabled=enabled)
    for i in range(8):
        rp.step(data[i % 4], data[(i + 1) % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rp.step(data[i % 4], data[(i + 1) % 4])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('%-9s %-10s %8.3f ms per iteration (host enqueue %7.3f)  %s' % (which, label, ms, t_host / steps * 1e3, rp.info()), flush=True)
    rp.invalidate()
    del model, rp
    torch.cuda.empty_cache()
