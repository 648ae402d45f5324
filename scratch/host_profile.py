"""Where the host (Python) time of an iteration goes: cProfile over enqueue-only iterations (no synchronisation inside).
python scratch/host_profile.py pix2pix|cyclegan|sagan|srgan [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs, attach_teacher

which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = {'pix2pix': 16, 'cyclegan': 1, 'sagan': 64, 'srgan': 16}[which]
common = ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)]
argv = {
    'pix2pix': ['--dataroot', 'synthetic', '--model', 'pix2pix', '--ngf', '32', '--ndf', '128', '--teacher_ngf', '64',
                '--teacher_ndf', '128', '--lambda_content', '50', '--lambda_gram', '1e4', '--arch_lr', '1e-4', '--arch_lr_step'],
    'cyclegan': ['--dataroot', 'synthetic', '--model', 'cyclegan', '--ngf', '24', '--ndf', '64', '--teacher_ngf', '64',
                 '--lambda_content', '0.01', '--lambda_gram', '10'],
    'sagan': ['--dataroot', 'synthetic', '--model', 'sagan', '--ngf', '48', '--ndf', '64', '--teacher_ngf', '64',
              '--crop_size', '64', '--gan_mode', 'hinge'],
    'srgan': ['--dataroot', 'synthetic', '--model', 'srgan', '--ngf', '24', '--teacher_ngf', '64', '--image_size', '96'],
}[which] + common
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
opt = options.parse(argv)
opt.isTrain = True
if not hasattr(opt, 'teacher_ndf') or opt.teacher_ndf is None:
    opt.teacher_ndf = opt.ndf
cls = get_model_class(opt)
model = cls(opt)
attach_teacher(model, opt, cls)
model.model_train()
data = [{k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in d.items()} for d in SyntheticPairs(opt, 4, 7)]


def step(i):
    model.set_input(data[i % 4])
    model.optimize_parameters()
    model.set_input(data[(i + 1) % 4])
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    step(i)
t_enq = (time.time() - t0) / steps * 1e3
torch.cuda.synchronize()
ms = (time.time() - t0) / steps * 1e3
print('%s batch %d: %.2f ms/step, host enqueue %.2f ms/step (unprofiled)' % (which, batch, ms, t_enq), flush=True)
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumtime'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    txt = s.getvalue()
    print('==== by %s (totals over %d steps) ====' % (key, steps))
    print('\n'.join(l[:170] for l in txt.splitlines()[4:60]))
