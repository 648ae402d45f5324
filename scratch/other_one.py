"""python scratch/other_one.py <cyclegan|sagan|srgan> [steps]: N iterations of one of bench.py's other configs (for rocprofv3:
sum of the kernel durations per iteration against the wall time per iteration = how launch-bound the model is)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gcc_amd import ops
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs, attach_teacher
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch, argv = bench.OTHER_ARGV[which]
opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
opt.isTrain = True
if getattr(opt, 'teacher_ndf', None) is None:
    opt.teacher_ndf = opt.ndf
cls = get_model_class(opt)
model = cls(opt)
attach_teacher(model, opt, cls)
model.model_train()
if os.environ.get('GCC_SERIALIZE') == '1' and hasattr(model, 'set_stream_schedule'):
    model.set_stream_schedule(False)          # every launch on one stream: a kernel's duration is its own
data = list(SyntheticPairs(opt, 4, 7))
def step(i):
    model.set_input(data[i % 4]); model.optimize_parameters()
    model.set_input(data[(i + 1) % 4]); model.clipping_mask_alpha(); model.optimizer_netD_arch()
for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print('%s: %.3f ms per iteration over %d (+5 warm-up) iterations' % (which, (time.perf_counter() - t0) / steps * 1e3, steps))
