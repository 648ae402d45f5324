"""python scratch/other_shapes.py <cyclegan|sagan|srgan|srgan_96_to_384>: every conv geometry of one bracketed single-stream iteration of
one of bench.py's other configs (launch count, average duration, TFLOP/s per shape), then the eager step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GCC_PROFILE_SHAPES'] = '1'
os.environ.setdefault('GCC_PROFILE_TOP', '60')
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
import torch
import bench
from gcc_amd import ops
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs, attach_teacher
which = sys.argv[1]
batch, argv = bench.OTHER_ARGV[which]
opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
opt.isTrain = True
if getattr(opt, 'teacher_ndf', None) is None:
    opt.teacher_ndf = opt.ndf
cls = get_model_class(opt)
model = cls(opt)
attach_teacher(model, opt, cls)
model.model_train()
data = [{k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in d.items()} for d in SyntheticPairs(opt, 4, 7)]
def step(i):
    model.set_input(data[i % 4]); model.optimize_parameters()
    model.set_input(data[(i + 1) % 4]); model.clipping_mask_alpha(); model.optimizer_netD_arch()
for i in range(5):
    step(i)
torch.cuda.synchronize()
ops.PROFILE.start(steps=1)
step(0)
ops.PROFILE.step_done()
r = ops.PROFILE.stop()
for k, v in sorted(r['per_kernel'].items(), key=lambda kv: -kv[1]['time_s']):
    print('%-16s %s' % (k, v))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    step(i)
torch.cuda.synchronize()
print('%s: %.3f ms per iteration' % (which, (time.perf_counter() - t0) / 10 * 1e3))
