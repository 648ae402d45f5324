"""Full-size soak of CycleGAN's two-sides fork (bench.py's configuration 3: 256 x 256, batch 1, distillation + architecture step):
`steps` iterations with GCC_CYCLE_FORK 0 / 1 / 2 must end on the same bits.   python scratch/soak_cyclegan_forks.py [steps]"""
import hashlib, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GCC_VGG19_RANDOM', '1')
import torch
import bench
from gcc_amd.models import CycleGAN as Cg, get_model_class
from gcc_amd.options import options
from gcc_amd.train import SyntheticPairs, attach_teacher

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def digest(model):
    h = hashlib.sha256()
    for m in (model, model.teacher_model):
        for k, v in sorted(m.state_dict().items()):
            h.update(k.encode()); h.update(v.detach().float().cpu().numpy().tobytes())
        for name in sorted(dir(m)):
            o = getattr(m, name, None)
            for pn in ('plan', 'plan_dup'):
                p = getattr(o, pn, None) if name.startswith('optimizer') else None
                if p is not None:
                    for a, b in zip(p.m, p.v):
                        h.update(a.detach().cpu().numpy().tobytes()); h.update(b.detach().cpu().numpy().tobytes())
    return h.hexdigest()


out = {}
for mode in (0, 1, 2):
    Cg.CYCLE_FORK = mode
    random.seed(5); torch.manual_seed(5)
    batch, argv = bench.OTHER_ARGV['cyclegan']
    opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
    opt.isTrain = True
    if getattr(opt, 'teacher_ndf', None) is None:
        opt.teacher_ndf = opt.ndf
    cls = get_model_class(opt)
    model = cls(opt)
    attach_teacher(model, opt, cls)
    model.model_train()
    data = [{k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in d.items()} for d in SyntheticPairs(opt, 4, 7)]
    for i in range(steps):
        model.set_input(data[i % 4]); model.optimize_parameters()
        model.set_input(data[(i + 1) % 4]); model.clipping_mask_alpha(); model.optimizer_netD_arch()
    torch.cuda.synchronize()
    out[mode] = digest(model)
    print('GCC_CYCLE_FORK=%d: %s  %s' % (mode, out[mode][:16], {k: round(v, 5) for k, v in list(model.get_current_losses().items())[:6]}), flush=True)
    del model
    torch.cuda.empty_cache()
print('IDENTICAL' if len(set(out.values())) == 1 else 'DIFFERENT')
sys.exit(0 if len(set(out.values())) == 1 else 1)
