#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REAL REFERENCE.

Runs only in the authoring container (needs /root/reference; never runs on the GPU box, never
imported by tests).  The reference's Python is imported unmodified with empty stand-ins for the
packages this image lacks (torchvision / cv2 / thop / skimage -- SURVEY.md Appendix B); nothing from
the reference is copied: the outputs are data (inputs + expected outputs).

    python tests/golden/make_fixtures.py            # rewrites tests/golden/*.npz

Fixtures
  (weights marked "recipe" are NOT stored: tests/golden/recipe.py regenerates them; the script
   loads the recipe values into the reference's modules before running them)
  pix2pix_eval_d8.npz   eval-mode U-Net (num_downs 8, ngf 8, 256x256), recipe weights: input, fake_B
  pix2pix_gcc_d6.npz    student(ngf8,ndf8,masked D)+teacher(ngf16,ndf16), num_downs 6, 64x64, N=2,
                        --no_dropout, direction BtoA, recipe weights: 2 x (optimize_parameters +
                        arch step): inputs, per-iteration losses, iteration-1 hooked features /
                        targets / fake_B, final state (small tensors whole, large ones subsampled
                        at recipe.sample_idx positions; BN running stats, alpha included)
  pix2pix_pretrain_d6.npz  plain Pix2Pix (no teacher, plain D, lambda_scale 1e-2): 2 iterations
  ops.npz               DifferentiableOP fwd/bwd (alpha <,==,> tau), GANLoss x4 modes, gram,
                        LambdaLR values, init_weights statistics
  options.json          options.parse() results for 7 command lines (flag surface + per-model overrides)
  pix2pix_pruned_d8.npz pruned student built from filter_cfgs/channel_cfgs with irregular widths: eval image + 1 iteration
  pix2pix_resnet_gcc.npz  --backbone resnet (MobileResnet + InstanceNorm) GCC iteration: eval/train images, features, losses
  cyclegan_gcc.npz      MobileCycleGAN student + online teacher, 2 x (optimize_parameters + arch step): images, features, losses, final state
  cyclegan_pretrain.npz CycleGAN without teacher, --lambda_weight (heavy-layer L1 sparsity), 1 iteration; ImagePool(3) sequence
  prune_resnet.npz      resnet_prune / CycleGAN get_prunenet_cfg cfgs + max_min_conv_norm; pruned MobileResnet (one with a block removed): eval + 1 iteration
  sagan_gcc.npz         SAGAN student + online teacher (spectral norm, self attention, duplicated optimizer entries): eval image, 2 x (iteration + arch step)
  srgan_gcc.npz         SRGAN student + online teacher (SRResNet, avg-pool discriminator, VGG stand-in): eval image, 2 x (iteration + arch step)
  prune_search_d8.npz   binarysearch_threshold trajectory end points with a documented thop stand-in
  prune_d8.npz          scale_prune / norm_prune cfgs + max_min_* at several thresholds (ngf 8)
"""
import copy
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from recipe import recipe_state_dict, recipe_transform, sample_idx, srgan_condition, shape_bn_scales_for_removal  # noqa: E402
REF = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms')
    _stub('torchvision.transforms.functional')
    _stub('torchvision.utils', make_grid=None)
    _stub('torchvision.models')
    _stub('torchvision.datasets')
    _stub('torchvision.models.vgg', vgg19=None)
    _stub('cv2', INTER_AREA=3)
    _stub('thop', profile=None)
    _stub('skimage')
    _stub('skimage.metrics', peak_signal_noise_ratio=None, structural_similarity=None)
    sys.path.insert(0, REF)


def parse(argv):
    from options import options
    sys.argv = ['train.py'] + argv
    opt = options.parse()
    opt.isTrain = True
    return opt


def sd_np(prefix, sd, out):
    for k, v in sd.items():
        out[prefix + k] = v.detach().cpu().numpy().copy()


def build_gcc(opt):
    """train.py:84-105"""
    from models import get_model_class
    cls = get_model_class(opt)
    model = cls(opt)
    topt = copy.deepcopy(opt)
    topt.ngf = opt.teacher_ngf
    topt.ndf = opt.teacher_ndf
    topt.darts_discriminator = False
    topt.online_distillation = False
    topt.generator_only = False
    teacher = cls(topt)
    teacher.model_train()
    setattr(model, 'teacher_model', teacher)
    model.init_distillation()
    teacher.init_distillation()
    return model, teacher


def load_recipe(module, seed):
    sd = module.state_dict()
    rec = recipe_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict(rec)


def sd_np_sampled(prefix, sd, out):
    for k, v in sd.items():
        v = v.detach().cpu().reshape(-1)
        out[prefix + k] = v[sample_idx(v.numel())].numpy().copy()


def fixture_eval_d8():
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1',
                 '--ngf', '8', '--ndf', '8', '--no_dropout'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 31)
    model.model_eval()
    out = {}
    g = torch.Generator().manual_seed(5)
    A = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    B = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    with torch.no_grad():
        model.forward()
    out['A'] = A.numpy()
    out['B'] = B.numpy()
    out['direction'] = np.array(opt.direction)
    out['seed_G'] = np.array(31)
    out['fake_B'] = model.get_current_visuals()['fake_B'].numpy()
    np.savez_compressed(os.path.join(HERE, 'pix2pix_eval_d8.npz'), **out)
    print('pix2pix_eval_d8: fake_B', out['fake_B'].shape, 'direction', opt.direction)


def fixture_gcc_d6():
    torch.manual_seed(7)
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1',
                 '--ngf', '8', '--ndf', '8', '--teacher_ngf', '16', '--num_downs', '6', '--no_dropout',
                 '--online_distillation', '--darts_discriminator', '--lambda_content', '50',
                 '--lambda_gram', '1e4', '--arch_lr', '1e-4', '--arch_lr_step'])
    opt.teacher_ndf = 16           # parse() forces 128; shrunk so the fixture stays small
    opt.batch_size = 2
    model, teacher = build_gcc(opt)
    model.model_train()
    out = {'direction': np.array(opt.direction), 'threshold': np.array(opt.threshold),
           'flags': np.array(' '.join(sys.argv[1:]) + ' (teacher_ndf=16)'),
           'seeds': np.array([101, 102, 103, 104, 105])}     # sG sD tG tD T
    load_recipe(model.netG, 101)
    load_recipe(model.netD, 102)
    load_recipe(teacher.netG, 103)
    load_recipe(teacher.netD, 104)
    with torch.no_grad():
        for i, t in enumerate(model.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 105 + i))
        # nudge alphas so that the gates have <, == and > tau channels
        a = model.netD.model[2].alpha
        a[0] = 0.2
        a[1] = 0.5
        model.netD.model[5].alpha[3] = 0.1
    for k, v in model.netD.state_dict().items():
        if k.endswith('alpha'):
            out['init.sD.' + k] = v.numpy().copy()
    g = torch.Generator().manual_seed(99)
    n_iter = 2
    for it in range(n_iter):
        A = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        B = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        vA = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        vB = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        out['it%d.A' % it], out['it%d.B' % it] = A.numpy(), B.numpy()
        out['it%d.vA' % it], out['it%d.vB' % it] = vA.numpy(), vB.numpy()
        model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        model.optimize_parameters()
        if it == 0:
            out['it0.fake_B'] = model.fake_B.detach().numpy().copy()
            out['it0.Tfake_B'] = teacher.fake_B.detach().numpy().copy()
            for j, f in enumerate(model.target_distillation_features):
                out['it0.target.%d' % j] = f.detach().numpy().copy()
            for j, f in enumerate(model.get_distillation_features()):
                out['it0.sfeat.%d' % j] = f.detach().numpy().copy()
            for j, f in enumerate(teacher.total_discriminator_features.values()):
                out['it0.tDfeat_on_sfake.%d' % j] = f.detach().numpy().copy()
            sd_np_sampled('it0.afterstep.sG.', model.netG.state_dict(), out)
            sd_np_sampled('it0.afterstep.sD.', model.netD.state_dict(), out)
        model.set_input({'A': vA, 'B': vB, 'A_paths': ['a'], 'B_paths': ['b']})
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
        for k, v in model.get_current_losses().items():
            out['it%d.loss.%s' % (it, k)] = np.array(v, dtype=np.float64)
        for k in ('G_GAN', 'G_L1', 'D_real', 'D_fake'):
            out['it%d.tloss.%s' % (it, k)] = np.array(float(getattr(teacher, 'loss_' + k)), dtype=np.float64)
    sd_np_sampled('final.sG.', model.netG.state_dict(), out)
    sd_np_sampled('final.sD.', model.netD.state_dict(), out)
    sd_np_sampled('final.tG.', teacher.netG.state_dict(), out)
    sd_np_sampled('final.tD.', teacher.netD.state_dict(), out)
    for i, t in enumerate(model.transform_convs):
        out['final.T.%d' % i] = t.weight.detach().numpy().copy()
    out['threads'] = np.array(torch.get_num_threads())
    np.savez_compressed(os.path.join(HERE, 'pix2pix_gcc_d6.npz'), **out)
    print('pix2pix_gcc_d6:', {k: float(out[k]) for k in out if k.startswith('it1.loss')})


def fixture_pretrain_d6():
    torch.manual_seed(21)
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1',
                 '--ngf', '8', '--ndf', '8', '--num_downs', '6', '--no_dropout', '--lambda_scale', '1e-2'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    model.model_train()
    out = {'direction': np.array(opt.direction), 'seeds': np.array([201, 202])}
    load_recipe(model.netG, 201)
    load_recipe(model.netD, 202)
    g = torch.Generator().manual_seed(3)
    for it in range(2):
        A = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        B = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
        out['it%d.A' % it], out['it%d.B' % it] = A.numpy(), B.numpy()
        model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        model.optimize_parameters()
        for k, v in model.get_current_losses().items():
            out['it%d.loss.%s' % (it, k)] = np.array(v, dtype=np.float64)
    sd_np_sampled('final.G.', model.netG.state_dict(), out)
    sd_np_sampled('final.D.', model.netD.state_dict(), out)
    # LR schedule values as the reference's scheduler produces them (n_epochs 10, decay 15 here)
    lrs = []
    for ep in range(1, opt.n_epochs + opt.n_epochs_decay + 1):
        model.update_learning_rate(ep)
        lrs.append(model.optimizers[0].param_groups[0]['lr'])
    out['lr_after_epoch'] = np.array(lrs, dtype=np.float64)
    out['sched'] = np.array([opt.epoch_count, opt.n_epochs, opt.n_epochs_decay, opt.lr], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'pix2pix_pretrain_d6.npz'), **out)
    print('pix2pix_pretrain_d6 ok; n_epochs', opt.n_epochs, opt.n_epochs_decay)


def fixture_ops():
    from models.DifferentiableOp import DifferentiableOP
    from models.GANLoss import GANLoss
    from models import get_model_class
    import utils.util as util
    out = {}
    g = torch.Generator().manual_seed(17)
    # gate
    op = DifferentiableOP(6, 0.5)
    with torch.no_grad():
        op.alpha.copy_(torch.tensor([0.1, 0.5, 0.9, 0.5000001, 0.4999999, 1.0]))
    x = torch.randn(3, 6, 5, 4, generator=g, requires_grad=True)
    dy = torch.randn(3, 6, 5, 4, generator=g)
    y = op(x)
    y.backward(dy)
    out['gate.x'], out['gate.dy'], out['gate.alpha'] = x.detach().numpy(), dy.numpy(), op.alpha.detach().numpy()
    out['gate.y'], out['gate.dx'], out['gate.dalpha'] = y.detach().numpy(), x.grad.numpy(), op.alpha.grad.numpy()
    out['gate.mask'] = op.get_current_mask().detach().numpy()
    # gan losses
    pred = torch.randn(2, 1, 6, 6, generator=g) * 1.5
    out['gan.pred'] = pred.numpy()
    for mode in ('hinge', 'lsgan', 'vanilla', 'wgangp'):
        crit = GANLoss(mode)
        for real in (True, False):
            for ford in (True, False):
                if mode == 'hinge' and not ford and not real:
                    continue
                p = pred.clone().requires_grad_(True)
                l = crit(p, real, for_discriminator=ford)
                l.backward()
                key = 'gan.%s.%d.%d' % (mode, int(real), int(ford))
                out[key] = np.array(float(l), dtype=np.float64)
                out[key + '.grad'] = p.grad.numpy()
    # gram (method of the model; build a tiny one)
    opt = parse(['--dataroot', './database/x/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '4',
                 '--ndf', '4', '--num_downs', '6'])
    torch.manual_seed(1)
    model = get_model_class(opt)(opt)
    f = torch.randn(2, 5, 7, 3, generator=g)
    out['gram.x'] = f.numpy()
    out['gram.y'] = model.gram(f).numpy()
    # init_weights statistics (seeded; compared statistically)
    w = []
    bn_w, bn_b = [], []
    for m in model.netG.modules():
        cn = m.__class__.__name__
        if cn.find('Conv') != -1:
            w.append(m.weight.detach().flatten())
        elif cn.find('BatchNorm2d') != -1:
            bn_w.append(m.weight.detach().flatten())
            bn_b.append(m.bias.detach().flatten())
    w, bn_w, bn_b = torch.cat(w), torch.cat(bn_w), torch.cat(bn_b)
    out['init.stats'] = np.array([float(w.mean()), float(w.std()), float(bn_w.mean()), float(bn_w.std()),
                                  float(bn_b.mean()), float(bn_b.std())])
    out['init.G_keys'] = np.array(list(model.netG.state_dict().keys()))
    out['init.D_keys'] = np.array(list(model.netD.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)
    print('ops ok')


def fixture_prune_d8():
    torch.manual_seed(0)
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1',
                 '--ngf', '8', '--ndf', '4', '--scale_prune'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    out = {'seed_G': np.array(301)}
    load_recipe(model.netG, 301)
    with torch.no_grad():      # spread the BN scales so that thresholds prune different amounts
        gsp = torch.Generator().manual_seed(302)
        for m in model.netG.modules():
            if m.__class__.__name__ == 'BatchNorm2d':
                m.weight.copy_(1.0 + 0.02 * torch.randn(m.weight.shape, generator=gsp))
    mx, mn = model.max_min_bn_scale()
    out['bn.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
    ths = [float(mn) - 0.01, 0.97, 0.99, 1.0, 1.01, 1.03, float(mx)]
    out['bn.thresholds'] = np.array(ths, dtype=np.float64)
    for i, t in enumerate(ths):
        pm = model.scale_prune(t)
        f, c = pm.get_cfg()
        out['bn.f.%d' % i], out['bn.c.%d' % i] = np.array(f), np.array(c)
    opt2 = copy.deepcopy(opt)
    opt2.scale_prune, opt2.norm_prune = False, True
    model.opt = opt2
    mx, mn = model.max_min_conv_norm()
    out['norm.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
    ths = [float(mn) * 0.5, float(mn) + 0.3 * (float(mx) - float(mn)), float(mn) + 0.6 * (float(mx) - float(mn)),
           float(mx) * 0.999]
    out['norm.thresholds'] = np.array(ths, dtype=np.float64)
    for i, t in enumerate(ths):
        pm = model.norm_prune(t)
        f, c = pm.get_cfg()
        out['norm.f.%d' % i], out['norm.c.%d' % i] = np.array(f), np.array(c)
    np.savez_compressed(os.path.join(HERE, 'prune_d8.npz'), **out)
    print('prune_d8 ok: bn cfg@1.0 =', list(out['bn.f.3']))


def fixture_pruned_d8():
    """pruned student (irregular widths from scale_prune at threshold 1.0 of the prune_d8 model): eval image and
    one plain training iteration (recipe weights, seed 401/402)"""
    z = np.load(os.path.join(HERE, 'prune_d8.npz'))
    f, c = [int(v) for v in z['bn.f.3']], [int(v) for v in z['bn.c.3']]
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                 '--no_dropout'])
    from models import get_model_class
    model = get_model_class(opt)(opt, filter_cfgs=f, channel_cfgs=c)
    load_recipe(model.netG, 401)
    load_recipe(model.netD, 402)
    out = {'f': np.array(f), 'c': np.array(c), 'seeds': np.array([401, 402]), 'direction': np.array(opt.direction)}
    g = torch.Generator().manual_seed(8)
    A = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    B = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    out['A'], out['B'] = A.numpy(), B.numpy()
    model.model_eval()
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    with torch.no_grad():
        model.forward()
    out['eval.fake_B'] = model.fake_B.numpy().copy()
    model.model_train()
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    model.optimize_parameters()
    out['train.fake_B'] = model.fake_B.detach().numpy().copy()
    for k, v in model.get_current_losses().items():
        out['loss.%s' % k] = np.array(v, dtype=np.float64)
    sd_np_sampled('final.G.', model.netG.state_dict(), out)
    np.savez_compressed(os.path.join(HERE, 'pix2pix_pruned_d8.npz'), **out)
    print('pix2pix_pruned_d8 ok: f =', f)


def fixture_pruned_removed_d8():
    """Pruned students that LOST inner blocks (models/Pix2Pix.py:87, 97, Identity submodule :59-67).  A full ngf-32 U-Net with
    BatchNorm scales shaped by recipe.shape_bn_scales_for_removal is searched by the reference's binarysearch_threshold (thop
    stand-in above) for two budgets -> block 7 gone / blocks 6 and 7 gone; a direct scale_prune(0.96) removes 5, 6 and 7.  Each
    pruned student then gets recipe weights and runs an eval image and one plain training iteration; the 6-block student also
    runs one full GCC iteration (distillation from an ngf-48 teacher + arch step) with its hooked features."""
    import utils.prune_util as pu
    pu.profile = thop_standin_profile
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '32', '--ndf', '4',
                 '--scale_prune', '--no_dropout'])
    from models import get_model_class
    cls = get_model_class(opt)
    model = cls(opt)
    load_recipe(model.netG, 311)
    shape_bn_scales_for_removal(model.netG.state_dict(), 312)       # state_dict tensors alias the parameters
    full, _ = pu.get_flops_parms(model.netG, model.device, opt)
    mx, mn = model.max_min_bn_scale()
    out = {'full_macs': np.array(full), 'seeds': np.array([311, 312]), 'max_min': np.array([float(mx), float(mn)], dtype=np.float64),
           'direction': np.array(opt.direction)}
    # budgets = what the thresholds 0.58 (block 7 gone) and 0.88 (6 and 7 gone, depth 5 thinned) cost, so that the search has
    # an answer there; it stops at the first mid point within 0.1 G of the budget
    cfgs = {}
    for tag, probe in (('k7', 0.58), ('k6', 0.88)):
        pm = model.scale_prune(probe)
        target, _ = pu.get_flops_parms(pm.netG, pm.device, opt)
        target = round(target, 3)
        thr = pu.binarysearch_threshold(model, target)
        pm = model.prune(thr)
        f, c = pm.get_cfg()
        macs, _ = pu.get_flops_parms(pm.netG, pm.device, opt)
        out[tag + '.target'], out[tag + '.threshold'] = np.array(target), np.array(float(thr), dtype=np.float32)
        out[tag + '.f'], out[tag + '.c'], out[tag + '.macs'] = np.array(f), np.array(c), np.array(macs)
        cfgs[tag] = (f, c)
    pm = model.scale_prune(0.96)
    f, c = pm.get_cfg()
    out['k5.threshold'], out['k5.f'], out['k5.c'] = np.array(0.96, dtype=np.float32), np.array(f), np.array(c)
    out['k5.macs'] = np.array(pu.get_flops_parms(pm.netG, pm.device, opt)[0])
    cfgs['k5'] = (f, c)
    g = torch.Generator().manual_seed(9)
    A = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    B = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    out['A'], out['B'] = A.numpy(), B.numpy()
    popt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '32', '--ndf', '8',
                  '--no_dropout'])
    for i, tag in enumerate(('k7', 'k6', 'k5')):
        f, c = cfgs[tag]
        pm = cls(popt, filter_cfgs=f, channel_cfgs=c)
        load_recipe(pm.netG, 411 + 2 * i)
        load_recipe(pm.netD, 412 + 2 * i)
        out[tag + '.keys'] = np.array(list(pm.netG.state_dict().keys()))
        pm.model_eval()
        pm.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        with torch.no_grad():
            pm.forward()
        out[tag + '.eval.fake_B'] = pm.fake_B.numpy()[:, :, ::2, ::2].copy()
        pm.model_train()
        pm.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        pm.optimize_parameters()
        out[tag + '.train.fake_B'] = pm.fake_B.detach().numpy()[:, :, ::2, ::2].copy()
        for k, v in pm.get_current_losses().items():
            out[tag + '.loss.%s' % k] = np.array(v, dtype=np.float64)
        sd_np_sampled(tag + '.final.G.', pm.netG.state_dict(), out)
    # one GCC iteration of the 6-block student
    f, c = cfgs['k6']
    gopt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '32', '--ndf', '8',
                  '--teacher_ngf', '48', '--no_dropout', '--online_distillation', '--darts_discriminator', '--lambda_content', '50',
                  '--lambda_gram', '1e4', '--arch_lr', '1e-4', '--arch_lr_step'])
    gopt.teacher_ndf = 8
    student = cls(gopt, filter_cfgs=f, channel_cfgs=c)
    topt = copy.deepcopy(gopt)
    topt.ngf, topt.ndf, topt.darts_discriminator, topt.online_distillation, topt.generator_only = 48, 8, False, False, False
    teacher = cls(topt)
    teacher.model_train()
    student.teacher_model = teacher
    student.init_distillation()
    teacher.init_distillation()
    student.model_train()
    for m, sd in ((student.netG, 421), (student.netD, 422), (teacher.netG, 423), (teacher.netD, 424)):
        load_recipe(m, sd)
    with torch.no_grad():
        for i, t in enumerate(student.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 425 + i))
    vA = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    vB = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    out['vA'], out['vB'] = vA.numpy(), vB.numpy()
    student.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    student.optimize_parameters()
    out['gcc.fake_B'] = student.fake_B.detach().numpy()[:, :, ::2, ::2].copy()
    for j, t in enumerate(student.get_distillation_features()):
        t = t.detach().reshape(-1)
        out['gcc.sfeat.%d' % j] = t[sample_idx(t.numel(), 8192)].numpy().copy()
        out['gcc.sfeat_shape.%d' % j] = np.array(student.get_distillation_features()[j].shape)
    student.set_input({'A': vA, 'B': vB, 'A_paths': ['a'], 'B_paths': ['b']})
    student.clipping_mask_alpha()
    student.optimizer_netD_arch()
    for k, v in student.get_current_losses().items():
        out['gcc.loss.%s' % k] = np.array(v, dtype=np.float64)
    sd_np_sampled('gcc.final.sG.', student.netG.state_dict(), out)
    for i, t in enumerate(student.transform_convs):
        t = t.weight.detach().reshape(-1)
        out['gcc.final.T.%d' % i] = t[sample_idx(t.numel())].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'pix2pix_pruned_removed_d8.npz'), **out)
    print('pruned_removed_d8 ok:', {t: (float(out[t + '.threshold']), list(out[t + '.f'])) for t in ('k7', 'k6', 'k5')})


def thop_standin_profile(model, inputs, verbose=False):
    """Stand-in for thop.profile (thop is not installed here and the reference does not pin a version): counts,
    with forward hooks on a real forward pass, Conv2d / ConvTranspose2d as out_elements * (Cin/groups) * kh * kw
    and BatchNorm2d as 2 * elements -- the convention SURVEY.md section 8(c) documents for the reference's budgets."""
    import torch.nn as nn
    total = [0]
    hooks = []

    def conv_hook(m, i, o):
        total[0] += o.numel() * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]

    def bn_hook(m, i, o):
        total[0] += 2 * i[0].numel()
    def softmax_hook(m, i, o):          # thop's count_softmax: rows * (exp + add + div) = rows * (3 n - 1)
        n = i[0].size(m.dim)
        total[0] += (i[0].numel() // n) * (3 * n - 1)
    for m in model.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.BatchNorm2d):
            hooks.append(m.register_forward_hook(bn_hook))
        elif isinstance(m, nn.Softmax):
            hooks.append(m.register_forward_hook(softmax_hook))
    was = model.training
    model.eval()
    with torch.no_grad():
        model(*inputs)
    model.train(was)
    for h in hooks:
        h.remove()
    return float(total[0]), float(sum(p.numel() for p in model.parameters()))


def fixture_prune_search():
    """the reference's binarysearch_threshold (utils/prune_util.py:20-47) on the prune_d8 model, with the thop
    stand-in above: returned threshold (fp32) and the cfgs of model.prune(threshold) for three budgets"""
    import utils.prune_util as pu
    pu.profile = thop_standin_profile
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '4',
                 '--scale_prune'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 301)
    with torch.no_grad():
        gsp = torch.Generator().manual_seed(302)
        for m in model.netG.modules():
            if m.__class__.__name__ == 'BatchNorm2d':
                m.weight.copy_(1.0 + 0.02 * torch.randn(m.weight.shape, generator=gsp))
    full, _ = pu.get_flops_parms(model.netG, model.device, opt)
    out = {'full_macs': np.array(full), 'seed_G': np.array(301)}
    for i, frac in enumerate((0.45, 0.6, 0.8)):
        target = round(full * frac, 3)
        try:
            thr = pu.binarysearch_threshold(model, target)
            pm = model.prune(thr)
            f, c = pm.get_cfg()
            macs, _ = pu.get_flops_parms(pm.netG, pm.device, opt)
            out['s%d.found' % i] = np.array(1)
            out['s%d.threshold' % i] = np.array(float(thr), dtype=np.float32)
            out['s%d.f' % i], out['s%d.c' % i] = np.array(f), np.array(c)
            out['s%d.macs' % i] = np.array(macs)
        except NotImplementedError:
            out['s%d.found' % i] = np.array(0)
        out['s%d.target' % i] = np.array(target)
    np.savez_compressed(os.path.join(HERE, 'prune_search_d8.npz'), **out)
    print('prune_search ok', {k: (v.tolist() if v.size < 3 else '...') for k, v in out.items() if 'thr' in k or 'found' in k or 'macs' in k})


def fixture_resnet_gcc():
    """Pix2Pix with --backbone resnet (MobileResnetGenerator, InstanceNorm): student ngf 8 + masked D, online teacher
    ngf 16, 64x64, N=2: eval image, then one GCC iteration + arch step (recipe weights 501..505)"""
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--backbone', 'resnet',
                 '--ngf', '8', '--ndf', '8', '--teacher_ngf', '16', '--online_distillation', '--darts_discriminator',
                 '--lambda_content', '50', '--lambda_gram', '1e4', '--arch_lr', '1e-4', '--arch_lr_step'])
    opt.teacher_ndf = 16
    model, teacher = build_gcc(opt)
    load_recipe(model.netG, 501)
    load_recipe(model.netD, 502)
    load_recipe(teacher.netG, 503)
    load_recipe(teacher.netD, 504)
    with torch.no_grad():
        for i, t in enumerate(model.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 505 + i))
        model.netD.model[2].alpha[0] = 0.3
    out = {'direction': np.array(opt.direction), 'seeds': np.array([501, 502, 503, 504, 505]),
           'G_keys': np.array(list(model.netG.state_dict().keys()))}
    g = torch.Generator().manual_seed(77)
    A, B, vA, vB = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1 for _ in range(4))
    for n, t in (('A', A), ('B', B), ('vA', vA), ('vB', vB)):
        out[n] = t.numpy()
    model.model_eval()
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    with torch.no_grad():
        model.forward()
    out['eval.fake_B'] = model.fake_B.numpy().copy()
    model.model_train()
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    model.optimize_parameters()
    out['train.fake_B'] = model.fake_B.detach().numpy().copy()
    out['train.Tfake_B'] = teacher.fake_B.detach().numpy().copy()
    for j, f in enumerate(model.get_distillation_features()[:4]):
        out['sfeat.%d' % j] = f.detach().numpy().copy()
    for j, f in enumerate(model.target_distillation_features):
        out['target.%d' % j] = f.detach().numpy().copy()
    model.set_input({'A': vA, 'B': vB, 'A_paths': ['a'], 'B_paths': ['b']})
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()
    for k, v in model.get_current_losses().items():
        out['loss.%s' % k] = np.array(v, dtype=np.float64)
    sd_np_sampled('final.sG.', model.netG.state_dict(), out)
    sd_np_sampled('final.tG.', teacher.netG.state_dict(), out)
    sd_np_sampled('final.sD.', model.netD.state_dict(), out)
    np.savez_compressed(os.path.join(HERE, 'pix2pix_resnet_gcc.npz'), **out)
    print('pix2pix_resnet_gcc ok', {k: round(float(v), 4) for k, v in out.items() if k.startswith('loss.')})


CYCLE_ARGV = ['--dataroot', './database/horse2zebra/', '--model', 'cyclegan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
              '--teacher_ngf', '16', '--online_distillation', '--darts_discriminator', '--lambda_content', '0.01',
              '--lambda_gram', '10', '--arch_lr', '1e-4', '--arch_lr_step']


def fixture_cyclegan():
    """MobileCycleGANModel student (ngf 8, masked BN discriminators ndf 8) + online teacher (ngf 16, InstanceNorm
    discriminators ndf 16), 64x64, N=2, recipe weights 601..: two iterations of optimize_parameters + arch step."""
    opt = parse(CYCLE_ARGV)
    opt.teacher_ndf = 16
    model, teacher = build_gcc(opt)
    nets = [(model.netG_A, 601), (model.netG_B, 602), (model.netD_A, 603), (model.netD_B, 604),
            (teacher.netG_A, 605), (teacher.netG_B, 606), (teacher.netD_A, 607), (teacher.netD_B, 608)]
    for n, sd in nets:
        load_recipe(n, sd)
    with torch.no_grad():
        for i, t in enumerate(model.transform_A_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 620 + i))
        for i, t in enumerate(model.transform_B_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 630 + i))
        model.netD_A.model[2].alpha[0] = 0.3
        model.netD_B.model[5].alpha[1] = 0.45
    out = {'direction': np.array(opt.direction), 'gan_mode': np.array(opt.gan_mode),
           'lambda_L1': np.array(opt.lambda_L1), 'lambda_A': np.array(opt.lambda_A), 'lambda_B': np.array(opt.lambda_B),
           'lambda_identity': np.array(opt.lambda_identity),
           'D_keys': np.array(list(model.netD_A.state_dict().keys())),
           'TD_keys': np.array(list(teacher.netD_A.state_dict().keys())),
           'loss_names': np.array(model.loss_names), 'teacher_loss_names': np.array(teacher.loss_names)}
    g = torch.Generator().manual_seed(88)
    for it in range(2):
        A, B, vA, vB = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1 for _ in range(4))
        for n, t in (('A', A), ('B', B), ('vA', vA), ('vB', vB)):
            out['it%d.%s' % (it, n)] = t.numpy()
        model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        model.optimize_parameters()
        if it == 0:
            for n in ('fake_A', 'fake_B', 'rec_A', 'rec_B', 'idt_A', 'idt_B'):
                out['it0.' + n] = getattr(model, n).detach().numpy().copy()
            out['it0.Tfake_A'] = teacher.fake_A.detach().numpy().copy()
            out['it0.Tfake_B'] = teacher.fake_B.detach().numpy().copy()
            for w, tg in (('A', model.target_distillation_A_features), ('B', model.target_distillation_B_features)):
                for j, f in enumerate(tg):
                    out['it0.target_%s.%d' % (w, j)] = f.detach().numpy().copy()
            for w in 'AB':
                for j, f in enumerate(model.get_distillation_features(AorB=w)[:4]):
                    out['it0.sfeat_%s.%d' % (w, j)] = f.detach().numpy().copy()
        model.set_input({'A': vA, 'B': vB, 'A_paths': ['a'], 'B_paths': ['b']})
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
        for k, v in model.get_current_losses().items():
            out['it%d.loss.%s' % (it, k)] = np.array(v, dtype=np.float64)
        for k, v in teacher.get_current_losses().items():
            out['it%d.tloss.%s' % (it, k)] = np.array(v, dtype=np.float64)
    for tag, net in (('sG_A', model.netG_A), ('sG_B', model.netG_B), ('sD_A', model.netD_A), ('sD_B', model.netD_B),
                     ('tG_A', teacher.netG_A), ('tG_B', teacher.netG_B), ('tD_A', teacher.netD_A), ('tD_B', teacher.netD_B)):
        sd_np_sampled('final.%s.' % tag, net.state_dict(), out)
    for w, tc in (('A', model.transform_A_convs), ('B', model.transform_B_convs)):
        for i, t in enumerate(tc):
            out['final.T_%s.%d' % (w, i)] = t.weight.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'cyclegan_gcc.npz'), **out)
    print('cyclegan_gcc ok', {k: round(float(v), 4) for k, v in out.items() if k.startswith('it1.loss.')})


def fixture_cyclegan_pretrain():
    """pretrain_for_pruning configuration: no teacher, InstanceNorm discriminators, --lambda_weight (L1 sparsity with the
    heavy-layer multipliers): one iteration; plus the reference ImagePool on a full pool with Python's random seeded."""
    import random
    opt = parse(['--dataroot', './database/horse2zebra/', '--model', 'cyclegan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                 '--lambda_weight', '1e-3'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    for n, sd in ((model.netG_A, 641), (model.netG_B, 642), (model.netD_A, 643), (model.netD_B, 644)):
        load_recipe(n, sd)
    g = torch.Generator().manual_seed(89)
    A, B = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1 for _ in range(2))
    out = {'A': A.numpy(), 'B': B.numpy(), 'direction': np.array(opt.direction)}
    model.model_train()
    model.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
    model.optimize_parameters()
    for k, v in model.get_current_losses().items():
        out['loss.%s' % k] = np.array(v, dtype=np.float64)
    for tag, net in (('G_A', model.netG_A), ('G_B', model.netG_B), ('D_A', model.netD_A), ('D_B', model.netD_B)):
        sd_np_sampled('final.%s.' % tag, net.state_dict(), out)
    from utils.image_pool import ImagePool
    random.seed(1234)
    pool = ImagePool(3)
    seq = []
    for step in range(8):
        imgs = torch.arange(2, dtype=torch.float32).reshape(2, 1, 1, 1) + 10 * step
        seq.append(pool.query(imgs).reshape(-1).numpy().copy())
    out['pool.returned'] = np.stack(seq)
    np.savez_compressed(os.path.join(HERE, 'cyclegan_pretrain.npz'), **out)
    print('cyclegan_pretrain ok', {k: round(float(v), 4) for k, v in out.items() if k.startswith('loss.')}, out['pool.returned'].tolist())


def spread_filter_norms(net, seed):
    """scale every conv filter by a seeded factor in [0.5, 1.5) so that norm thresholds prune different amounts"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if m.__class__.__name__ in ('Conv2d', 'ConvTranspose2d'):
                f = 0.5 + torch.rand(m.weight.shape[0], generator=g)
                m.weight.mul_(f.reshape(-1, 1, 1, 1))


def fixture_prune_resnet():
    """resnet_prune (Pix2Pix, union-of-masks residual rule) and CycleGAN get_prunenet_cfg (mean-norm residual rule) +
    their max_min_conv_norm at several thresholds; a pruned MobileResnet (irregular widths, and one with a residual
    block removed): eval image and one plain training iteration"""
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--backbone', 'resnet',
                 '--ngf', '8', '--ndf', '8', '--norm_prune'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 701)
    spread_filter_norms(model.netG, 702)
    out = {}
    mx, mn = model.max_min_conv_norm()
    out['p2p.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
    ths = [float(mn) * 0.5] + [float(mn) + q * (float(mx) - float(mn)) for q in (0.2, 0.5, 0.8, 0.999)]
    out['p2p.thresholds'] = np.array(ths, dtype=np.float64)
    for i, t in enumerate(ths):
        out['p2p.f.%d' % i] = np.array(model.resnet_prune(t).get_cfg()[0])
    copt = parse(['--dataroot', './database/horse2zebra/', '--model', 'cyclegan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                  '--norm_prune'])
    cm = get_model_class(copt)(copt)
    load_recipe(cm.netG_A, 703)
    spread_filter_norms(cm.netG_A, 704)
    mx, mn = cm.max_min_conv_norm(cm.netG_A)
    out['cyc.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
    ths = [float(mn) * 0.5] + [float(mn) + q * (float(mx) - float(mn)) for q in (0.2, 0.5, 0.8, 0.999)]
    out['cyc.thresholds'] = np.array(ths, dtype=np.float64)
    for i, t in enumerate(ths):
        out['cyc.f.%d' % i] = np.array(cm.get_prunenet_cfg(cm.netG_A, t))
    # pruned generators through the Pix2Pix step
    g = torch.Generator().manual_seed(9)
    A = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    B = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    out['A'], out['B'], out['direction'] = A.numpy(), B.numpy(), np.array(opt.direction)
    cfg_a = [int(v) for v in out['p2p.f.2']]
    cfg_b = list(cfg_a)
    cfg_b[5] = 0                      # residual block 2 removed (its mid width is 0): later Sequential indices shift
    for tag, cfg in (('a', cfg_a), ('b', cfg_b)):
        popt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--backbone', 'resnet',
                      '--ngf', '8', '--ndf', '8'])
        pm = get_model_class(popt)(popt, filter_cfgs=cfg)
        load_recipe(pm.netG, 711)
        load_recipe(pm.netD, 712)
        out['pruned_%s.cfg' % tag] = np.array(cfg)
        out['pruned_%s.G_keys' % tag] = np.array(list(pm.netG.state_dict().keys()))
        pm.model_eval()
        pm.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        with torch.no_grad():
            pm.forward()
        out['pruned_%s.eval.fake_B' % tag] = pm.fake_B.numpy().copy()
        pm.model_train()
        pm.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        pm.optimize_parameters()
        for k, v in pm.get_current_losses().items():
            out['pruned_%s.loss.%s' % (tag, k)] = np.array(v, dtype=np.float64)
        sd_np_sampled('pruned_%s.final.G.' % tag, pm.netG.state_dict(), out)
    np.savez_compressed(os.path.join(HERE, 'prune_resnet.npz'), **out)
    print('prune_resnet ok:', list(out['p2p.f.2']), list(out['cyc.f.2']))


def fixture_sagan():
    """SAGANModel student (ngf 8, masked D ndf 8) + online teacher (ngf 16, ndf 16), z_dim 128, 64x64, N=4, recipe
    weights 801..: eval image, then two iterations of optimize_parameters + arch step.  torch >= 2 rejects the
    reference's Adam(betas=(0, 0.9)) (int 0): the script casts betas to float on the way in (SURVEY.md hazard H7)."""
    import torch.optim as optim
    real_adam = optim.Adam

    class FloatBetasAdam(real_adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatBetasAdam
    try:
        opt = parse(['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                     '--teacher_ngf', '16', '--online_distillation', '--darts_discriminator', '--threshold', '0.1',
                     '--lambda_L1', '1', '--lambda_content', '1', '--lambda_gram', '1', '--arch_lr', '1e-4'])
        opt.teacher_ndf = 16
        model, teacher = build_gcc(opt)
    finally:
        torch.optim.Adam = real_adam
    for n, sd in ((model.netG, 801), (model.netD, 802), (teacher.netG, 803), (teacher.netD, 804)):
        load_recipe(n, sd)
    with torch.no_grad():
        for i, t in enumerate(model.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 810 + i))
        model.netD.l1[1].alpha[0] = 0.3
        model.netD.l3[1].alpha[2] = 0.5
    out = {'G_keys': np.array(list(model.netG.state_dict().keys())), 'D_keys': np.array(list(model.netD.state_dict().keys())),
           'TD_keys': np.array(list(teacher.netD.state_dict().keys())), 'gan_mode': np.array(opt.gan_mode),
           'lr': np.array(opt.lr), 'batch_size': np.array(opt.batch_size), 'crop_size': np.array(opt.crop_size),
           'loss_names': np.array(model.loss_names),
           'T_shapes': np.array([list(t.weight.shape[:2]) for t in model.transform_convs])}
    for tag, optim_ in (('G', model.optimizer_G), ('D', model.optimizer_D)):
        ids = [id(p) for p in optim_.param_groups[0]['params']]
        names = {id(p): k for k, p in list(model.netG.named_parameters()) + list(model.netD.named_parameters())}
        out['dup_' + tag] = np.array(sorted({names[i] for i in ids if ids.count(i) > 1 and i in names}))
    g = torch.Generator().manual_seed(90)
    z0 = torch.randn(4, 128, generator=g)
    model.model_eval()
    model.set_input({'z': z0, 'real_img': torch.zeros(4, 3, 64, 64), 'img_path': ['p'] * 4})
    sdG_before = {k: v.clone() for k, v in model.netG.state_dict().items()}
    with torch.no_grad():
        model.forward()
    out['eval.z'], out['eval.fake_img'] = z0.numpy(), model.fake_img.numpy().copy()
    model.netG.load_state_dict(sdG_before)          # the eval pass moved u, v: put them back
    model.model_train()
    for it in range(2):
        z, vz = torch.randn(4, 128, generator=g), torch.randn(4, 128, generator=g)
        real, vreal = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1 for _ in range(2))
        for n, t in (('z', z), ('vz', vz), ('real', real), ('vreal', vreal)):
            out['it%d.%s' % (it, n)] = t.numpy()
        model.set_input({'z': z, 'real_img': real, 'img_path': ['p'] * 4})
        model.optimize_parameters()
        if it == 0:
            out['it0.fake_img'] = model.fake_img.detach().numpy().copy()
            out['it0.Tfake_img'] = teacher.fake_img.detach().numpy().copy()
            for j, f in enumerate(model.target_distillation_features):
                out['it0.target.%d' % j] = f.detach().numpy().copy()
            for j, f in enumerate(model.get_distillation_features()[:2]):
                out['it0.sfeat.%d' % j] = f.detach().numpy().copy()
        model.set_input({'z': vz, 'real_img': vreal, 'img_path': ['p'] * 4})
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
        for k, v in model.get_current_losses().items():
            out['it%d.loss.%s' % (it, k)] = np.array(v, dtype=np.float64)
        for k, v in teacher.get_current_losses().items():
            out['it%d.tloss.%s' % (it, k)] = np.array(v, dtype=np.float64)
    for tag, net in (('sG', model.netG), ('sD', model.netD), ('tG', teacher.netG), ('tD', teacher.netD)):
        sd_np_sampled('final.%s.' % tag, net.state_dict(), out)
    for i, t in enumerate(model.transform_convs):
        out['final.T.%d' % i] = t.weight.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'sagan_gcc.npz'), **out)
    print('sagan_gcc ok', {k: round(float(v), 4) for k, v in out.items() if k.startswith('it1.loss.')})
    print('  G keys', list(out['G_keys'])[:14], '\n  D keys', list(out['D_keys']), '\n  dup G', list(out['dup_G']), '\n  dup D', list(out['dup_D']))


VGG_STANDIN = (8, 8, 'M', 16, 16, 'M', 32, 32, 32, 32, 'M', 64, 64, 64, 64, 'M', 64, 64, 64, 64, 'M')


def vgg19_standin(pretrained=True):
    """torchvision is not in the image: an nn.Module with the layer sequence of torchvision's vgg19().features (conv3 +
    ReLU(inplace) ... MaxPool2d(2, 2), 37 layers) at 1/8 of the widths and unset weights (the fixture loads recipe
    weights).  The reference's TruncatedVGG19 only iterates ``vgg.features.children()``."""
    import torch.nn as nn
    layers, cin = [], 3
    for c in VGG_STANDIN:
        if c == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, c, 3, padding=1), nn.ReLU(inplace=True)]
            cin = c
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


def fixture_srgan():
    """SRGAN student (ngf 8, masked D ndf 8) + online teacher (ngf 16, ndf 16), 12x12 -> 48x48, N=2, VGG stand-in,
    recipe weights 901..: eval image, two iterations of optimize_parameters + arch step.  The reference's parser does
    not declare --generator_only although parse() reads it (SURVEY.md hazard H7): the script adds the flag."""
    import torchvision.models.vgg as tvgg
    tvgg.vgg19 = vgg19_standin
    if 'PIL' not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            _stub('PIL', Image=None)
    from options import options as ref_options
    if not any('--generator_only' in a.option_strings for a in ref_options.parser._actions):
        ref_options.parser.add_argument('--generator_only', action='store_true')
    import models.GANLoss as ref_ganloss
    ref_ganloss.vgg19 = vgg19_standin
    opt = parse(['--dataroot', './database/sr/', '--model', 'srgan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                 '--teacher_ngf', '16', '--online_distillation', '--darts_discriminator', '--lambda_content', '1',
                 '--lambda_gram', '1', '--lambda_L1', '0.5', '--lambda_SR_content', '0.5', '--arch_lr', '1e-4'])
    opt.teacher_ndf = 16
    model, teacher = build_gcc(opt)
    for n, sd in ((model.netG, 901), (model.netD, 902), (teacher.netG, 903), (teacher.netD, 904),
                  (model.truncated_vgg19, 905)):
        load_recipe(n, sd)
    for n in (model.netG, model.netD, teacher.netG, teacher.netD):
        srgan_condition(n.state_dict())
    srgan_condition({'truncated_vgg19.' + k if not k.startswith('truncated_vgg19.') else k: v
                     for k, v in model.truncated_vgg19.state_dict().items()})
    teacher.truncated_vgg19.load_state_dict(model.truncated_vgg19.state_dict())
    with torch.no_grad():
        for i, t in enumerate(model.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 910 + i))
        model.netD.conv_blocks[0].conv_block[1].alpha[0] = 0.3
        model.netD.conv_blocks[2].conv_block[2].alpha[1] = 0.5
    out = {'G_keys': np.array(list(model.netG.state_dict().keys())), 'D_keys': np.array(list(model.netD.state_dict().keys())),
           'TD_keys': np.array(list(teacher.netD.state_dict().keys())),
           'V_keys': np.array(list(model.truncated_vgg19.state_dict().keys())),
           'gan_mode': np.array(opt.gan_mode), 'lr': np.array(opt.lr), 'threshold': np.array(opt.threshold),
           'loss_names': np.array(model.loss_names),
           'G_optimizer_names': np.array([k for k, p in model.netG.named_parameters()
                                          if any(p is q for q in model.optimizer_G.param_groups[0]['params'])])}
    g = torch.Generator().manual_seed(91)
    lr0 = torch.rand(2, 3, 12, 12, generator=g) * 2 - 1
    model.model_eval()
    model.set_input({'lr': lr0, 'hr': torch.zeros(2, 3, 48, 48), 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
    with torch.no_grad():
        model.forward()
    out['eval.lr'], out['eval.fake_hr'] = lr0.numpy(), model.fake_hr.numpy().copy()
    model.model_train()
    for it in range(2):
        lr_, vlr = (torch.rand(2, 3, 12, 12, generator=g) * 2 - 1 for _ in range(2))
        hr_, vhr = (torch.rand(2, 3, 48, 48, generator=g) * 2 - 1 for _ in range(2))
        for n, t in (('lr', lr_), ('hr', hr_), ('vlr', vlr), ('vhr', vhr)):
            out['it%d.%s' % (it, n)] = t.numpy()
        model.set_input({'lr': lr_, 'hr': hr_, 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
        model.optimize_parameters()
        if it == 0:
            out['it0.fake_hr_norm'] = model.fake_hr.detach().numpy().copy()       # ImageNet-normalised by backward_G
            for j, f in enumerate(model.target_distillation_features):
                out['it0.target.%d' % j] = f.detach().numpy().copy()
            for j, f in enumerate(model.get_distillation_features()[:4]):
                out['it0.sfeat.%d' % j] = f.detach().numpy().copy()
        model.set_input({'lr': vlr, 'hr': vhr, 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
        for k, v in model.get_current_losses().items():
            out['it%d.loss.%s' % (it, k)] = np.array(v, dtype=np.float64)
        for k, v in teacher.get_current_losses().items():
            out['it%d.tloss.%s' % (it, k)] = np.array(v, dtype=np.float64)
    for tag, net in (('sG', model.netG), ('sD', model.netD), ('tG', teacher.netG), ('tD', teacher.netD)):
        sd_np_sampled('final.%s.' % tag, net.state_dict(), out)
    for i, t in enumerate(model.transform_convs):
        out['final.T.%d' % i] = t.weight.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'srgan_gcc.npz'), **out)
    print('srgan_gcc ok', {k: round(float(v), 4) for k, v in out.items() if k.startswith('it1.loss.')})
    print('  loss names', list(out['loss_names']), len(out['G_optimizer_names']), len(out['G_keys']))


def fixture_srgan_content():
    """SRGAN.optimize_content_parameters (models/SRGAN.py:514-522): the generator-only MSE step with the BatchNorm-scale
    sparsity term (--lambda_scale), three iterations; no discriminator / VGG involved"""
    import torchvision.models.vgg as tvgg
    tvgg.vgg19 = vgg19_standin
    if 'PIL' not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            _stub('PIL', Image=None)
    from options import options as ref_options
    if not any('--generator_only' in a.option_strings for a in ref_options.parser._actions):
        ref_options.parser.add_argument('--generator_only', action='store_true')
    import models.GANLoss as ref_ganloss
    ref_ganloss.vgg19 = vgg19_standin
    opt = parse(['--dataroot', './database/sr/', '--model', 'srgan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                 '--generator_only', '--lambda_scale', '0.01'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 981)
    srgan_condition(model.netG.state_dict())
    model.model_train()
    g = torch.Generator().manual_seed(982)
    out = {'lr_G': np.array(opt.lr), 'loss_names': np.array(list(model.loss_names))}
    for it in range(3):
        lr_ = torch.rand(2, 3, 12, 12, generator=g) * 2 - 1
        hr_ = torch.rand(2, 3, 48, 48, generator=g) * 2 - 1
        out['it%d.lr' % it], out['it%d.hr' % it] = lr_.numpy(), hr_.numpy()
        model.set_input({'lr': lr_, 'hr': hr_, 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
        model.optimize_content_parameters()
        out['it%d.loss_content' % it] = np.array(float(model.loss_content), dtype=np.float64)
        if it == 0:
            out['it0.fake_hr'] = model.fake_hr.detach().numpy().copy()
    sd_np_sampled('final.G.', model.netG.state_dict(), out)
    np.savez_compressed(os.path.join(HERE, 'srgan_content.npz'), **out)
    print('srgan_content ok', [float(out['it%d.loss_content' % i]) for i in range(3)], list(out['loss_names']))


def _spread_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if m.__class__.__name__ == 'BatchNorm2d':
                m.weight.copy_(0.05 + torch.rand(m.weight.shape, generator=g))


def fixture_prune_search_gan():
    """the reference's own binarysearch_threshold + model.prune (utils/prune_util.py:20-63) for SRGAN (scale and norm
    pruning) and SAGAN (scale pruning), with the thop stand-in; also max_min_bn_scale / max_min_conv_norm"""
    import torchvision.models.vgg as tvgg
    tvgg.vgg19 = vgg19_standin
    if 'PIL' not in sys.modules:
        try:
            import PIL  # noqa: F401
        except ImportError:
            _stub('PIL', Image=None)
    from options import options as ref_options
    if not any('--generator_only' in a.option_strings for a in ref_options.parser._actions):
        ref_options.parser.add_argument('--generator_only', action='store_true')
    import models.GANLoss as ref_ganloss
    ref_ganloss.vgg19 = vgg19_standin
    import utils.prune_util as pu
    pu.profile = thop_standin_profile
    from models import get_model_class
    out = {}
    for tag, flag in (('sr_scale', '--scale_prune'), ('sr_norm', '--norm_prune')):
        opt = parse(['--dataroot', './database/sr/', '--model', 'srgan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8', flag])
        model = get_model_class(opt)(opt)
        load_recipe(model.netG, 951)
        _spread_bn(model.netG, 952)
        spread_filter_norms(model.netG, 953)
        full, _ = pu.get_flops_parms(model.netG, model.device, opt)
        out[tag + '.full_macs'] = np.array(full)
        mx, mn = model.max_min_bn_scale() if flag == '--scale_prune' else model.max_min_conv_norm()
        out[tag + '.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
        for i, frac in enumerate((0.7, 0.8, 0.9)):
            target = round(full * frac, 4)
            out['%s.s%d.target' % (tag, i)] = np.array(target)
            try:
                thr = pu.binarysearch_threshold(model, target)
                pm = model.prune(thr)
                macs, _ = pu.get_flops_parms(pm.netG, pm.device, opt)
                out['%s.s%d.found' % (tag, i)] = np.array(1)
                out['%s.s%d.threshold' % (tag, i)] = np.array(float(thr), dtype=np.float32)
                out['%s.s%d.f' % (tag, i)] = np.array(pm.get_cfg()[0])
                out['%s.s%d.macs' % (tag, i)] = np.array(macs)
            except NotImplementedError:
                out['%s.s%d.found' % (tag, i)] = np.array(0)
    # torch >= 2.x rejects the reference's mixed int / float betas (0, 0.9): same shim as fixture_sagan, kept installed
    # for the whole search (model.prune builds a new SAGANModel at every mid point)
    real_adam = torch.optim.Adam

    class FloatBetasAdam(real_adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatBetasAdam
    # ngf 32: a pruned l3 / l4 must keep >= 8 channels, or Self_Attn's in_dim // 8 query conv has no filters
    opt = parse(['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '-1', '--ngf', '32', '--ndf', '8',
                 '--scale_prune'])
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 961)
    _spread_bn(model.netG, 962)
    full, _ = pu.get_flops_parms(model.netG, model.device, opt)
    out['sa.full_macs'] = np.array(full)
    mx, mn = model.max_min_bn_scale()
    out['sa.max_min'] = np.array([float(mx), float(mn)], dtype=np.float64)
    for i, frac in enumerate((0.5, 0.65, 0.8)):
        target = round(full * frac, 4)
        out['sa.s%d.target' % i] = np.array(target)
        try:
            thr = pu.binarysearch_threshold(model, target)
            pm = model.prune(thr)
            macs, _ = pu.get_flops_parms(pm.netG, pm.device, opt)
            out['sa.s%d.found' % i] = np.array(1)
            out['sa.s%d.threshold' % i] = np.array(float(thr), dtype=np.float32)
            out['sa.s%d.f' % i] = np.array(pm.get_cfg()[0])
            out['sa.s%d.macs' % i] = np.array(macs)
        except NotImplementedError:
            out['sa.s%d.found' % i] = np.array(0)
    torch.optim.Adam = real_adam
    np.savez_compressed(os.path.join(HERE, 'prune_search_gan.npz'), **out)
    print('prune_search_gan ok', {k: v.tolist() for k, v in out.items() if v.size < 20})


def fixture_checkpoint():
    """a checkpoint file written by the reference's Pix2PixModel.save_models (masked D, ngf 4 / ndf 4, num_downs 8 at
    256x256 would be 3 MB: the generator here is the d6 one of fixture_gcc_d6 at ngf 4), and the eval image the
    reference produces from it"""
    import tempfile
    opt = parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1', '--ngf', '4', '--ndf', '4',
                 '--num_downs', '6', '--load_size', '64', '--crop_size', '64', '--darts_discriminator'])
    from models import get_model_class
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 971)
    load_recipe(model.netD, 972)
    with tempfile.TemporaryDirectory() as d:
        model.save_models(7, d, fid=12.5)
        blob = open(os.path.join(d, 'model_7.pth'), 'rb').read()
    open(os.path.join(HERE, 'ref_checkpoint_pix2pix.pth'), 'wb').write(blob)
    g = torch.Generator().manual_seed(973)
    A = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    model.model_eval()
    model.set_input({'A': A, 'B': A.clone(), 'A_paths': ['a'], 'B_paths': ['b']})
    with torch.no_grad():
        model.forward()
    np.savez_compressed(os.path.join(HERE, 'ref_checkpoint_pix2pix.npz'), A=A.numpy(), fake_B=model.fake_B.numpy(),
                        direction=np.array(opt.direction))
    print('checkpoint ok: %d bytes' % len(blob))


def fixture_checkpoint_other():
    """checkpoint files written by the reference's save_models of the other three model classes (student networks at the
    widths of their *_gcc fixtures, recipe weights), and the image each reference model produces from one in eval mode"""
    import tempfile
    import torch.optim as optim

    def dump(model, tag, fwd):
        with tempfile.TemporaryDirectory() as d:
            model.save_models(5, d, fid=7.25)
            blob = open(os.path.join(d, 'model_5.pth'), 'rb').read()
        open(os.path.join(HERE, 'ref_checkpoint_%s.pth' % tag), 'wb').write(blob)
        model.model_eval()
        with torch.no_grad():
            out = fwd(model)
        np.savez_compressed(os.path.join(HERE, 'ref_checkpoint_%s.npz' % tag), **{k: v.numpy() for k, v in out.items()})
        print('checkpoint %s ok: %d bytes' % (tag, len(blob)))

    # CycleGAN: two generators, two masked discriminators
    from models import get_model_class
    opt = parse([a for a in CYCLE_ARGV if a not in ('--online_distillation',)])
    model = get_model_class(opt)(opt)
    for n, sd in ((model.netG_A, 1101), (model.netG_B, 1102), (model.netD_A, 1103), (model.netD_B, 1104)):
        load_recipe(n, sd)
    g = torch.Generator().manual_seed(1105)
    A, B = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1, torch.rand(1, 3, 64, 64, generator=g) * 2 - 1

    def fwd_cycle(m):
        m.set_input({'A': A, 'B': B, 'A_paths': ['a'], 'B_paths': ['b']})
        m.forward()
        return {'A': A, 'B': B, 'fake_B': m.fake_B, 'fake_A': m.fake_A}
    dump(model, 'cyclegan', fwd_cycle)

    # SAGAN (torch >= 2 rejects the reference's integer beta: SURVEY.md hazard H7)
    real_adam = optim.Adam

    class FloatBetasAdam(real_adam):
        def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), **kw):
            super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), **kw)
    torch.optim.Adam = FloatBetasAdam
    try:
        opt = parse(['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                     '--darts_discriminator', '--threshold', '0.1'])
        model = get_model_class(opt)(opt)
    finally:
        torch.optim.Adam = real_adam
    load_recipe(model.netG, 1111)
    load_recipe(model.netD, 1112)
    z = torch.randn(4, opt.z_dim, generator=torch.Generator().manual_seed(1113))

    def fwd_sagan(m):
        m.set_input({'z': z, 'real_img': torch.zeros(4, 3, 64, 64), 'img_path': ['p'] * 4})
        m.forward()
        return {'z': z, 'fake_img': m.fake_img}
    dump(model, 'sagan', fwd_sagan)

    # SRGAN
    import torchvision.models.vgg as tvgg
    tvgg.vgg19 = vgg19_standin
    from options import options as ref_options
    if not any('--generator_only' in a.option_strings for a in ref_options.parser._actions):
        ref_options.parser.add_argument('--generator_only', action='store_true')
    import models.GANLoss as ref_ganloss
    ref_ganloss.vgg19 = vgg19_standin
    opt = parse(['--dataroot', './database/sr/', '--model', 'srgan', '--gpu_ids', '-1', '--ngf', '8', '--ndf', '8',
                 '--darts_discriminator'])
    model = get_model_class(opt)(opt)
    load_recipe(model.netG, 1121)
    load_recipe(model.netD, 1122)
    srgan_condition(model.netG.state_dict())
    srgan_condition(model.netD.state_dict())
    lr = torch.rand(2, 3, 12, 12, generator=torch.Generator().manual_seed(1123))

    def fwd_srgan(m):
        m.set_input({'lr': lr, 'hr': torch.zeros(2, 3, 48, 48), 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
        m.forward()
        return {'lr': lr, 'fake_hr': m.fake_hr}
    dump(model, 'srgan', fwd_srgan)


def fixture_metric():
    """evaluation arithmetic of the reference on seeded inputs: calculate_frechet_distance (scipy sqrtm) for a
    well-conditioned and a rank-deficient pair, np.mean / np.cov statistics, fast_hist + per_class_iu, y-channel
    conversion"""
    # import stubs only: metric/fid_score.py:42 imports cv2.imread, metric/inception.py subclasses
    # torchvision's Inception blocks at import time; the arithmetic below never touches either
    sys.modules['cv2'].imread = None
    base = type('InceptionBlock', (torch.nn.Module,), {})
    sys.modules['torchvision.models'].inception = _stub('torchvision.models.inception', InceptionA=base, InceptionC=base,
                                                        InceptionE=base)
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    from metric.fid_score import calculate_frechet_distance
    rng = np.random.RandomState(77)
    out = {}
    for tag, d, n1, n2 in (('full', 48, 400, 300), ('wide', 96, 700, 500), ('rank', 40, 25, 30)):
        basis = rng.randn(d, d) / np.sqrt(d)
        a1 = (rng.randn(n1, d) * (0.2 + rng.rand(d))) @ basis + rng.randn(d) * 0.3
        a2 = (rng.randn(n2, d) * (0.2 + rng.rand(d))) @ basis.T + rng.randn(d) * 0.3
        a1, a2 = a1.astype(np.float32), a2.astype(np.float32)
        m1, s1 = np.mean(a1.astype(np.float64), axis=0), np.cov(a1.astype(np.float64), rowvar=False)
        m2, s2 = np.mean(a2.astype(np.float64), axis=0), np.cov(a2.astype(np.float64), rowvar=False)
        out['%s.act1' % tag], out['%s.act2' % tag] = a1, a2
        out['%s.mu1' % tag], out['%s.sigma1' % tag] = m1, s1
        out['%s.fid' % tag] = np.array(float(calculate_frechet_distance(m1, s1, m2, s2)))
    from metric.mIoU_score import fast_hist, per_class_iu
    n = 19
    label = rng.randint(-1, n + 3, size=4000).astype(np.int64)      # includes ignored labels (< 0, >= n; 255 in cityscapes)
    label[rng.rand(label.size) < 0.05] = 255
    scores = rng.randn(2, n, 40, 50).astype(np.float32)
    scores[0, 3, 5, 7] = scores[0, 9, 5, 7] = scores[0].max() + 1          # a tie: the first maximum wins
    pred = scores.argmax(axis=1).reshape(-1)
    hist = fast_hist(pred, label, n)
    out['iou.scores'], out['iou.label'], out['iou.pred'] = scores, label, pred.astype(np.int64)
    out['iou.hist'] = hist.astype(np.int64)
    out['iou.per_class'] = per_class_iu(hist.astype(np.float64))
    out['iou.miou'] = np.array(round(np.nanmean(per_class_iu(hist.astype(np.float64)) * 100), 2))
    from data.sr_dataset import convert_image
    g = torch.Generator().manual_seed(78)
    fake = torch.rand(2, 3, 40, 36, generator=g) * 2 - 1
    real = (fake + 0.1 * torch.randn(2, 3, 40, 36, generator=g)).clamp(-1, 1)
    out['psnr.fake'], out['psnr.real'] = fake.numpy(), real.numpy()
    out['psnr.fake_y'] = convert_image(fake, source='[-1, 1]', target='y-channel').numpy()
    out['psnr.real_y'] = convert_image(real, source='[-1, 1]', target='y-channel').numpy()
    np.savez_compressed(os.path.join(HERE, 'metric.npz'), **out)
    print('metric ok', {k: float(v) for k, v in out.items() if v.size == 1})


def fixture_pipeline():
    """the paired-image transform chain of data/aligned_dataset.py:40-53 with PIL 12.2.0 (torchvision absent: its
    Resize / crop / flip / ToTensor / Normalize are spelled out with PIL + numpy), on seeded images: up- and
    down-scaling resizes, two crop / flip draws of the reference's get_params"""
    import random
    from PIL import Image
    from data.base_dataset import get_params
    rng = np.random.RandomState(91)
    out = {'pil_version': np.array(Image.__version__)}
    for tag, h, w, load, crop in (('up', 64, 64, 78, 64), ('down', 90, 70, 48, 40), ('same', 32, 32, 32, 32)):
        ab = (rng.rand(h, 2 * w, 3) * 255).astype(np.uint8)
        ab[: h // 2, : w // 3] = 255                                  # saturated patches: the clip of the fixed-point sums
        ab[h // 2:, w: w + w // 4] = 0
        out[tag + '.AB'] = ab
        opt = types.SimpleNamespace(preprocess='resize_and_crop', load_size=load, crop_size=crop, no_flip=False)
        random.seed(5 + h)
        for j in range(2):
            AB = Image.fromarray(ab).convert('RGB')
            ww, hh = AB.size
            w2 = int(ww / 2)
            A, B = AB.crop((0, 0, w2, hh)), AB.crop((w2, 0, ww, hh))
            p = get_params(opt, A.size)
            out['%s.%d.crop_pos' % (tag, j)], out['%s.%d.flip' % (tag, j)] = np.array(p['crop_pos']), np.array(p['flip'])
            for name, img in (('A', A), ('B', B)):
                img = img.resize((load, load), Image.BICUBIC)
                if j == 0:
                    out['%s.%s.resized' % (tag, name)] = np.array(img)
                x, y = p['crop_pos']
                if img.size[0] > crop or img.size[1] > crop:
                    img = img.crop((x, y, x + crop, y + crop))
                if p['flip']:
                    img = img.transpose(Image.FLIP_LEFT_RIGHT)
                t = torch.from_numpy(np.array(img)).permute(2, 0, 1).to(torch.float32).div(255)
                t = t.sub_(0.5).div_(0.5)
                out['%s.%d.%s' % (tag, j, name)] = t.numpy()
    np.savez_compressed(os.path.join(HERE, 'pipeline.npz'), **out)
    print('pipeline ok', {k: v.tolist() for k, v in out.items() if 'crop_pos' in k or 'flip' in k})


def fixture_options():
    import json
    from options import options
    cases = []
    for argv in (['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--ngf', '32', '--lambda_scale', '1e-2'],
                 ['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--ngf', '32', '--ndf', '128',
                  '--darts_discriminator', '--arch_lr', '1e-4', '--arch_lr_step', '--scale_prune', '--target_budget', '3.0',
                  '--online_distillation', '--lambda_content', '50', '--lambda_gram', '1e4', '--gpu_ids', '0'],
                 ['--dataroot', './database/maps', '--model', 'pix2pix'],
                 ['--dataroot', './database/edges2shoes-r', '--model', 'pix2pix', '--gpu_ids', '0,1'],
                 ['--dataroot', './database/horse2zebra', '--model', 'cyclegan', '--lambda_weight', '1e-3'],
                 ['--dataroot', './database/celeb', '--model', 'sagan'],
                 ['--dataroot', './database/church', '--model', 'sagan', '--gpu_ids', '-1']):
        sys.argv = ['train.py'] + argv
        parsed = vars(options.parse())
        parsed = {k: ('inf' if v == float('inf') else v) for k, v in parsed.items()}
        cases.append({'argv': argv, 'parsed': parsed})
    json.dump(cases, open(os.path.join(HERE, 'options.json'), 'w'), indent=1, sort_keys=True)
    print('options ok', len(cases))


if __name__ == '__main__':
    torch.set_num_threads(8)
    only = sys.argv[1:]            # e.g. "make_fixtures.py cyclegan cyclegan_pretrain"; none = all
    import_reference()
    for fn in (fixture_options, fixture_ops, fixture_eval_d8, fixture_gcc_d6, fixture_pretrain_d6, fixture_prune_d8,
               fixture_pruned_d8, fixture_pruned_removed_d8, fixture_prune_search, fixture_resnet_gcc, fixture_cyclegan, fixture_cyclegan_pretrain, fixture_prune_resnet, fixture_sagan, fixture_srgan, fixture_prune_search_gan, fixture_checkpoint, fixture_checkpoint_other, fixture_metric, fixture_pipeline, fixture_srgan_content):
        if not only or fn.__name__[len('fixture_'):] in only:
            fn()
