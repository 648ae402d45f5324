"""Step time of the other model families (BASELINE.json configs 2-4) on synthetic batches, teacher stream on / off.
python scratch/bench_models.py cyclegan|sagan|srgan [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.models import get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs, attach_teacher

which = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else {'cyclegan': 1, 'sagan': 64, 'srgan': 16}[which]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
common = ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)]
argv = {
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
data = list(SyntheticPairs(opt, 4, 7))


def step(i):
    model.set_input(data[i % 4])
    model.optimize_parameters()
    model.set_input(data[(i + 1) % 4])
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()


for serial in (True, False, True, False):
    model.serialize_streams = serial
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        step(i)
    t_enq = (time.time() - t0) / steps * 1e3
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    print('%s batch %d  %s: %.2f ms/step (host enqueue %.2f ms)  %.1f img/s' % (
        which, batch, 'one stream' if serial else 'teacher stream', ms, t_enq, batch / ms * 1e3), flush=True)
