"""SAGAN on the HIP path: spectral-norm and attention kernels against plain PyTorch fp32, the model against the
reference's golden vectors (tests/golden/sagan_gcc.npz) and the oracle's gradients.  Tolerances as in
tests/test_pix2pix_gpu.py unless stated."""
import copy
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_pix2pix_gpu import DEV, _rel, load, load_recipe

pytestmark = pytest.mark.gpu


def _rb(t):
    return t.bfloat16().float()


def _to_nhwc(ops, x):
    buf = ops.new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], DEV)
    ops.nchw_to_nhwc(x.to(DEV).contiguous(), buf)
    return buf


@pytest.mark.parametrize('shape,transposed', [((32, 16, 4, 4), False), ((128, 24, 4, 4), True), ((8, 3, 4, 4), False)])
def test_spectral_norm_power_iteration_and_gradient(shape, transposed):
    """one power iteration, sigma, W_bar / sigma and the gradient fold against autograd (fp32 on both sides)"""
    from gcc_amd import ops
    g = torch.Generator().manual_seed(shape[0])
    w = torch.randn(shape, generator=g) * 0.1
    u0, v0 = torch.randn(shape[0], generator=g), torch.randn(shape[1] * 16, generator=g)
    G = torch.randn(shape, generator=g)
    # reference arithmetic (models/SAGAN.py:25-38) with autograd through sigma
    wr = w.clone().requires_grad_(True)
    wm = wr.reshape(shape[0], -1)
    v = torch.mv(wm.detach().t(), u0)
    v = v / (v.norm() + 1e-12)
    t = torch.mv(wm.detach(), v)
    u = (t / (t.norm() + 1e-12)).requires_grad_(True)
    vv = v.clone().requires_grad_(True)
    sigma = u.dot(wm.mv(vv))
    ((wr / sigma) * G).sum().backward()
    # device: the master is channels_last like a FlatParams-homed conv weight
    wd = w.to(DEV).contiguous(memory_format=torch.channels_last)
    ud, vd = u0.to(DEV), v0.to(DEV)
    t_d, s_d = torch.zeros(shape[0], device=DEV), torch.zeros(1, device=DEV)
    w_eff = torch.empty_like(wd)
    ops.spectral_power_iteration(wd, ud, vd, t_d, s_d, w_eff)
    torch.cuda.synchronize()
    assert torch.allclose(ud.cpu(), u.detach(), atol=1e-5) and torch.allclose(vd.cpu(), v, atol=1e-5)
    assert abs(s_d.item() - sigma.item()) <= 1e-5 * abs(sigma.item())
    assert torch.allclose(w_eff.cpu(), (w / sigma.detach()), atol=1e-5)
    Gd = G.to(DEV).contiguous(memory_format=torch.channels_last)
    dw, du, dv = torch.zeros_like(wd), torch.zeros_like(ud), torch.zeros_like(vd)
    ops.spectral_grad(Gd, wd, ud, vd, t_d, s_d, dw, du=du, dv=dv)
    torch.cuda.synchronize()
    assert _rel(dw.cpu(), wr.grad) <= 1e-4, _rel(dw.cpu(), wr.grad)
    assert _rel(du.cpu(), u.grad) <= 1e-4 and _rel(dv.cpu(), vv.grad) <= 1e-4


@pytest.mark.parametrize('shape', [(32, 16, 4, 4), (128, 24, 4, 4), (8, 3, 4, 4), (100, 52, 3, 3), (48, 48, 1, 1), (1, 384, 4, 4)])
def test_spectral_power_iteration_fused_with_the_packings(shape):
    """gcc_spectral_power_iteration_pack (W_bar / sigma straight into the two bf16 packings: 4 launches) against the separate route
    (gcc_spectral_power_iteration + gcc_pack_weights: 7): u, v, t, sigma and both packings bit for bit, padding included"""
    from gcc_amd import ops
    R, Cc, k, _ = shape
    g = torch.Generator().manual_seed(R + Cc)
    w = (torch.randn(shape, generator=g) * 0.1).to(DEV)
    wd = w.contiguous(memory_format=torch.channels_last) if k > 1 else w.contiguous()
    u0, v0 = torch.randn(R, generator=g).to(DEV), torch.randn(Cc * k * k, generator=g).to(DEV)
    outs = []
    for fused in (False, True):
        u, v = u0.clone(), v0.clone()
        t, s = torch.zeros(R, device=DEV), torch.zeros(1, device=DEV)
        pw = torch.full((ops.ceil8(R), k * k, ops.ceil8(Cc)), 7.0, dtype=torch.bfloat16, device=DEV)
        pwt = torch.full((ops.ceil8(Cc), k * k, ops.ceil8(R)), 7.0, dtype=torch.bfloat16, device=DEV)
        pw[R:] = 0                                   # rows beyond R are the allocation's zeros on both routes (never written)
        ops.lib().gcc_launch_count(1)
        if fused:
            ops.spectral_power_iteration_pack(wd, u, v, t, s, pw, pwt)
        else:
            w_eff = torch.empty_like(wd)
            ops.spectral_power_iteration(wd, u, v, t, s, w_eff)
            ops.pack_weights_into(w_eff, pw, pwt)
        n = int(ops.lib().gcc_launch_count(1))
        torch.cuda.synchronize()
        outs.append((u, v, t, s, pw, pwt, n))
    a, b = outs
    for x, y, what in zip(a[:6], b[:6], ('u', 'v', 't', 'sigma', 'W', 'Wt')):
        assert torch.equal(x, y), what
    if ops.ceil8(Cc) > Cc:
        assert float(b[4][:R, :, Cc:].float().abs().max()) == 0.0      # padding columns of the rows that exist: zeros
    if ops.ceil8(R) > R:
        assert float(b[5][:Cc, :, R:].float().abs().max()) == 0.0
    assert b[6] == 4 and a[6] >= 6, (a[6], b[6])


def test_grouped_spectral_power_iterations_are_the_single_calls_bit_for_bit():
    """gcc_spectral_power_iteration_pack_group (the power iterations of all spectrally normalised layers of a forward pass: four
    launches) against one gcc_spectral_power_iteration_pack per layer: u, v, t, sigma and both packings of every layer bit for
    bit -- eleven layers (two groups of <= 8), the shapes of the single-call test plus SAGAN's own"""
    from gcc_amd import ops, _lib
    shapes = [(32, 16, 4, 4), (128, 24, 4, 4), (8, 3, 4, 4), (100, 52, 3, 3), (48, 48, 1, 1), (1, 384, 4, 4),
              (128, 512, 4, 4), (512, 256, 4, 4), (256, 128, 4, 4), (64, 3, 4, 4), (128, 64, 4, 4)]
    assert len(shapes) > _lib.SPECTRAL_GROUP_MAX
    g = torch.Generator().manual_seed(5)
    layers = []
    for shape in shapes:
        R, Cc, k, _ = shape
        w = (torch.randn(shape, generator=g) * 0.1).to(DEV)
        wd = w.contiguous(memory_format=torch.channels_last) if k > 1 else w.contiguous()
        layers.append((wd, torch.randn(R, generator=g).to(DEV), torch.randn(Cc * k * k, generator=g).to(DEV)))

    def fresh():
        out = []
        for wd, u0, v0 in layers:
            R, Cc, k, _ = wd.shape
            pw = torch.full((ops.ceil8(R), k * k, ops.ceil8(Cc)), 7.0, dtype=torch.bfloat16, device=DEV)
            pwt = torch.full((ops.ceil8(Cc), k * k, ops.ceil8(R)), 7.0, dtype=torch.bfloat16, device=DEV)
            pw[R:] = 0
            out.append((wd, u0.clone(), v0.clone(), torch.zeros(R, device=DEV), torch.zeros(1, device=DEV), pw, pwt))
        return out
    single, grouped = fresh(), fresh()
    for e in single:
        ops.spectral_power_iteration_pack(*e)
    ops.lib().gcc_launch_count(1)
    for rep in range(2):                             # (a second iteration on the moved u, v: the scratch of a group is reused)
        if rep:
            for e in single:
                ops.spectral_power_iteration_pack(*e)
            ops.lib().gcc_launch_count(1)
        ops.spectral_power_iteration_pack_group(grouped)
        n = int(ops.lib().gcc_launch_count(1))
        torch.cuda.synchronize()
        assert n == 8, n
        for a, b, shape in zip(single, grouped, shapes):
            for x, y, what in zip(a[1:], b[1:], ('u', 'v', 't', 'sigma', 'W', 'Wt')):
                assert torch.equal(x, y), (shape, what, rep)


@pytest.mark.parametrize('B,C,H', [(2, 64, 16), (3, 16, 8), (2, 512, 4), (1, 48, 32), (2, 96, 12), (2, 256, 8), (1, 64, 7),
                                   (2, 8, 32), (2, 32, 8), (1, 24, 9)])
def test_self_attention_forward_backward(B, C, H):
    """y = gamma * softmax(q^T k) v + x and its gradients w.r.t. q, k, v, gamma (q, k with C // 8 channels)"""
    from gcc_amd import ops
    g = torch.Generator().manual_seed(C + H)
    C8, N = C // 8, H * H
    q, k = _rb(torch.randn(B, C8, H, H, generator=g)), _rb(torch.randn(B, C8, H, H, generator=g))
    v, x = _rb(torch.randn(B, C, H, H, generator=g)), _rb(torch.randn(B, C, H, H, generator=g))
    dy = _rb(torch.randn(B, C, H, H, generator=g))
    gamma = torch.tensor([0.7])
    qr, kr, vr, gr = (t.clone().requires_grad_(True) for t in (q, k, v, gamma))
    attn = torch.softmax(torch.bmm(qr.reshape(B, C8, N).permute(0, 2, 1), kr.reshape(B, C8, N)), dim=-1)
    o = torch.bmm(vr.reshape(B, C, N), attn.permute(0, 2, 1)).reshape(B, C, H, H)
    y_ref = gr * o + x
    (y_ref * dy).sum().backward()
    c8p = ops.ceil8(C8)
    offs = (0, c8p, 2 * c8p)
    qkv = ops.new_act(B, 2 * c8p + C, H, H, DEV)
    for t, off in ((q, 0), (k, c8p), (v, 2 * c8p)):
        ops.nhwc_copy(_to_nhwc(ops, t), 0, qkv, off, t.shape[1])
    xd = _to_nhwc(ops, x)
    y, od = ops.new_act(B, C, H, H, DEV), ops.new_act(B, C, H, H, DEV)
    A = torch.zeros((B, N, N), device=DEV)
    stats = torch.zeros((B, N, 2), device=DEV)
    gd = gamma.to(DEV)
    ops.attention_fwd(qkv, offs, xd, gd, C, C8, y, od, stats, A=A)
    torch.cuda.synchronize()
    assert torch.allclose(A.cpu(), attn.detach(), atol=2e-6 + 1e-5 * float(attn.max()))
    energy = torch.bmm(q.reshape(B, C8, N).permute(0, 2, 1), k.reshape(B, C8, N))
    assert torch.allclose(stats[..., 0].cpu(), energy.max(-1).values, atol=1e-4, rtol=1e-5)
    assert _rel(ops.nhwc_to_nchw(y, C).cpu(), y_ref.detach()) <= 5e-3
    y2, od2 = ops.new_act(B, C, H, H, DEV), ops.new_act(B, C, H, H, DEV)
    ops.attention_fwd(qkv, offs, xd, gd, C, C8, y2, od2, torch.zeros_like(stats))       # the model's call: no map
    assert torch.equal(y2, y) and torch.equal(od2, od)
    dqkv = ops.new_act(B, 2 * c8p + C, H, H, DEV)
    rowdot = torch.zeros((B, N), device=DEV)
    dgam = torch.zeros(1, device=DEV)
    ops.attention_bwd(qkv, offs, od, stats, gd, _to_nhwc(ops, dy), C, C8, dqkv, rowdot, dgamma=dgam)
    torch.cuda.synchronize()
    dqkv2, dgam2 = ops.new_act(B, 2 * c8p + C, H, H, DEV), torch.zeros(1, device=DEV)
    ops.attention_bwd(qkv, offs, od, stats, gd, _to_nhwc(ops, dy), C, C8, dqkv2, rowdot, dgamma=dgam2)
    assert torch.equal(dqkv2, dqkv) and torch.equal(dgam2, dgam)          # no atomics: bit-reproducible
    full = ops.nhwc_to_nchw(dqkv, 2 * c8p + C).cpu()
    assert _rel(full[:, :C8], qr.grad) <= 1e-2, ('dq', _rel(full[:, :C8], qr.grad))
    assert _rel(full[:, c8p:c8p + C8], kr.grad) <= 1e-2, ('dk', _rel(full[:, c8p:c8p + C8], kr.grad))
    assert _rel(full[:, 2 * c8p:], vr.grad) <= 1e-2, ('dv', _rel(full[:, 2 * c8p:], vr.grad))
    # dgamma = sum dy o is a cancelling sum of B*C*N terms; o is stored in bf16 (and P enters the matrix cores in bf16):
    # three standard deviations of that rounding, 2^-9 per term
    noise = 3 * 2.0 ** -9 * float(((dy * o.detach()) ** 2).sum().sqrt())
    assert abs(dgam.item() - gr.grad.item()) <= 1e-2 * abs(gr.grad.item()) + 1e-3 + noise, (dgam.item(), gr.grad.item(), noise)


SAGAN_ARGV = ['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '0', '--ngf', '8', '--ndf', '8',
              '--teacher_ngf', '16', '--online_distillation', '--darts_discriminator', '--threshold', '0.1',
              '--lambda_L1', '1', '--lambda_content', '1', '--lambda_gram', '1', '--arch_lr', '1e-4']


def _build(z):
    from gcc_amd.options import options
    from gcc_amd.models import get_model_class
    from tests.golden.recipe import recipe_transform
    opt = options.parse(SAGAN_ARGV)
    opt.isTrain = True
    opt.teacher_ndf = 16
    cls = get_model_class(opt)
    model = cls(opt)
    topt = copy.deepcopy(opt)
    topt.ngf, topt.ndf = opt.teacher_ngf, opt.teacher_ndf
    topt.darts_discriminator = topt.online_distillation = False
    teacher = cls(topt)
    teacher.model_train()
    model.teacher_model = teacher
    model.init_distillation()
    teacher.init_distillation()
    for net, seed in ((model.netG, 801), (model.netD, 802), (teacher.netG, 803), (teacher.netD, 804)):
        load_recipe(net, seed)
    with torch.no_grad():
        for i, t in enumerate(model.transform_convs):
            t.weight.copy_(recipe_transform(t.weight.shape[0], t.weight.shape[1], 810 + i).to(DEV))
        model.netD.state_dict()['l1.1.alpha'][0] = 0.3
        model.netD.state_dict()['l3.1.alpha'][2] = 0.5
    model.refresh_weights()
    teacher.refresh_weights()
    model.model_train()
    return model, teacher, opt


def _batch(z, zk, rk):
    return {'z': torch.from_numpy(z[zk]), 'real_img': torch.from_numpy(z[rk]), 'img_path': ['p'] * 4}


_ZERO_G = lambda n: n.endswith('.module.bias') or n.endswith('key_conv.bias')
_ZERO_D = lambda n: n.endswith('key_conv.bias') or n == 'attn2.value_conv.bias'


def test_sagan_two_iterations_vs_reference_golden(golden_dir):
    from tests.golden.recipe import sample_idx
    z = load(golden_dir, 'sagan_gcc.npz')
    model, teacher, opt = _build(z)
    assert list(model.netG.state_dict().keys()) == [str(k) for k in z['G_keys']]
    assert list(model.netD.state_dict().keys()) == [str(k) for k in z['D_keys']]
    assert list(teacher.netD.state_dict().keys()) == [str(k) for k in z['TD_keys']]
    assert opt.gan_mode == str(z['gan_mode']) and abs(opt.lr - float(z['lr'])) < 1e-12
    assert model.loss_names == [str(k) for k in z['loss_names']]
    # eval image on the recipe state; then restore u, v (the eval pass moved them, as in the fixture script)
    sd0 = {k: v.clone() for k, v in model.netG.state_dict().items()}
    model.model_eval()
    model.set_input({'z': torch.from_numpy(z['eval.z']), 'real_img': torch.zeros(4, 3, 64, 64), 'img_path': ['p'] * 4})
    model.forward()
    e = (model.fake_img.cpu() - torch.from_numpy(z['eval.fake_img'])).abs()
    print('eval fake_img: max %.4g mean %.4g' % (e.max(), e.mean()))
    assert e.max() <= 2e-2 and e.mean() <= 3e-3
    model.netG.load_state_dict(sd0)
    model.model_train()
    from tests import _updates
    init = _updates.snapshot({'sG': model.netG, 'sD': model.netD, 'tG': teacher.netG, 'tD': teacher.netD})
    agree = _updates.MovementAgreement()
    masks = _updates.floor_masks(_oracle_grads(z, False), _oracle_grads(z, True))
    # the same two iterations on the oracle with bf16 storage emulated: the reference's arithmetic plus the rounding
    # points of the HIP path.  Behind sign-like Adam steps (beta1 = 0) the fp32 reference and any bf16 pipeline drift
    # apart (iteration 1: G_GAN 1.667 emulated vs 1.478 fp32), while the HIP path must stay on the emulated trajectory.
    from oracle import gcc_oracle as O
    from tests.test_oracle_golden import build_sagan_oracle
    emu = []
    # both errors of every scalar, apart (tests/_updates.LossBars): this family's asserts below are two-sided already (emulated
    # trajectory AND reference, each with its stated bar); the report names the scalars that sit within 3e-2 of the emulating oracle
    # only -- with beta1 = 0 the fixture's own emulation leaves the reference by up to 20 % in iteration 1 (tests/test_oracle_golden.py)
    bars = _updates.LossBars('sagan', n_map=10 ** 9)
    O.EMULATE_BF16 = True
    try:
        om, ot, _ = build_sagan_oracle(z)
        for it in range(2):
            om.set_input(torch.from_numpy(z['it%d.z' % it]), torch.from_numpy(z['it%d.real' % it]))
            om.optimize_parameters()
            om.set_input(torch.from_numpy(z['it%d.vz' % it]), torch.from_numpy(z['it%d.vreal' % it]))
            om.clipping_mask_alpha()
            om.optimizer_netD_arch()
            emu.append((dict(om.losses), dict(ot.losses)))
    finally:
        O.EMULATE_BF16 = False
    for it in range(2):
        model.set_input(_batch(z, 'it%d.z' % it, 'it%d.real' % it))
        model.optimize_parameters()
        if it == 0:
            e = (model.fake_img.cpu() - torch.from_numpy(z['it0.fake_img'])).abs()
            print('it0 fake_img: max %.4g mean %.4g' % (e.max(), e.mean()))
            assert e.max() <= 2e-2 and e.mean() <= 3e-3
            e = (teacher.fake_img.cpu() - torch.from_numpy(z['it0.Tfake_img'])).abs()
            assert e.max() <= 2e-2 and e.mean() <= 3e-3
            for j, f in enumerate(model.G.features(model._gctx)):
                ref = torch.from_numpy(z['it0.sfeat.%d' % j])
                err = (f.float().cpu() - ref).abs().max().item() / ref.abs().max().item()
                print('student feature %d: rel max err %.4g' % (j, err))
                assert err <= 3e-2
            for j in range(4):
                ref = torch.from_numpy(z['it0.target.%d' % j])
                err = (model.target_distillation_features[j].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
                print('target %d: rel max err %.4g' % (j, err))
                # target 3 is the teacher discriminator's attn2 feature computed AFTER its Adam step: with beta1 = 0 the
                # step is a pure sign step of 4e-4 per weight, so gradient elements whose sign bf16 rounding flips move the
                # 4x4 feature (oracle with bf16 storage emulated: 7.2% on this input)
                assert err <= (3e-2 if j < 3 else 0.12)
        model.set_input(_batch(z, 'it%d.vz' % it, 'it%d.vreal' % it))
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
        losses, tl = model.get_current_losses(), teacher.get_current_losses()
        for k in z.files:
            for pre, got, em in (('it%d.loss.' % it, losses, emu[it][0]), ('it%d.tloss.' % it, tl, emu[it][1])):
                if k.startswith(pre):
                    name, ref = k[len(pre):], float(z[k])
                    print('it%d %s %s: got %.5g  reference %.5g  bf16-emulating oracle %.5g' % (it, pre[-6], name, got[name], ref, em[name]))
                    # bar: 3e-2 of the emulated trajectory in iteration 0 (measured: 0.3%), 5e-2 in iteration 1, whose values
                    # sit behind a sign-like Adam step of every weight (measured: 4.1% on D_arch_diff, a hinge difference
                    # that the fp32 reference itself misses by 11% from the emulation); against the fp32 reference 3e-2
                    # before the first Adam step has acted (iteration 0: D_real, D_fake, content, gram, L1), else the
                    # measured drift 0.25
                    bar = 3e-2 if it == 0 else 5e-2
                    bars.add('it%d %s %s' % (it, pre[-6].replace('.', 'S'), name), name, got[name], ref, em[name])
                    assert abs(got[name] - em[name]) <= bar * max(1.0, abs(em[name])), (it, k, got[name], em[name])
                    pre_step = it == 0 and name in ('D_real', 'D_fake', 'content', 'gram', 'L1')
                    assert abs(got[name] - ref) <= (3e-2 if pre_step else 0.25) * max(1.0, abs(ref)), (it, k, got[name], ref)
    emu_sd = {'sG': om.G, 'sD': om.D, 'tG': ot.G, 'tD': ot.D}         # the emulating oracle after the same two iterations
    for tag, net, zero in (('sG', model.netG, _ZERO_G), ('sD', model.netD, _ZERO_D), ('tG', teacher.netG, _ZERO_G),
                           ('tD', teacher.netD, _ZERO_D)):
        sd = net.state_dict()
        prefix = 'final.%s.' % tag
        dup = tag in ('sG', 'sD')
        for k in z.files:
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if zero(name):
                continue
            ref = z[k]
            g = sd[name].detach().float().cpu().reshape(-1)
            g = g[sample_idx(g.numel())].numpy()
            if name.endswith('num_batches_tracked'):
                assert int(g[0]) == int(ref.reshape(-1)[0]), (tag, name)
                continue
            lr = opt.lr * (4 if 'D' in tag else 1)
            if name.endswith('running_mean') or name.endswith('running_var'):
                tol = 3e-2 * max(1.0, float(np.abs(ref).max()))
            elif name.endswith('alpha'):
                tol = 2.2 * opt.arch_lr * 2 + 1e-6
            elif name.endswith('weight_u') or name.endswith('weight_v'):
                tol = 8e-2            # unit vectors re-derived from W_bar by each power iteration (+ Adam steps in D): they
                                      # inherit the sign-step differences of W_bar, amplified by the matvec
            else:
                # Adam with beta1 = 0, beta2 = 0.9: update t moves a weight by at most lr * sqrt((1 - 0.9^t) / 0.1); a
                # flipped gradient sign doubles the distance.  2 steps, 4 updates for the duplicated entries.
                n_upd = 4 if (dup and ('.module.' in name or '_conv.' in name)) else 2
                tol = 2.2 * lr * sum(((1 - 0.9 ** t) / 0.1) ** 0.5 for t in range(1, n_upd + 1)) + 1e-6
            err = float(np.abs(g - ref).max())
            assert err <= tol, (tag, name, err, tol)
            if name.endswith('alpha'):
                agree.add(tag + '.alpha', init[tag][name], g, ref.reshape(-1), opt.arch_lr * 2, mask=masks.get(('alpha', name)),
                          emul=_updates.sampled(emu_sd[tag][name]))
            elif not (name.endswith('running_mean') or name.endswith('running_var') or name.endswith('weight_u')
                      or name.endswith('weight_v')):
                agree.add(tag, init[tag][name], g, ref.reshape(-1), lr * n_upd, mask=masks.get((tag, name)),
                          emul=_updates.sampled(emu_sd[tag][name]))
    bars.check(max_emul_only=0.35, require=False)     # measured on the CPU: 7 of this fixture's 24 emulated scalars sit > 3e-2 from the reference
    agree.check()


def _oracle_grads(z, emulate):
    """every parameter gradient of the first golden iteration + arch step on the oracle, learning rates 0 (fp32, or with bf16
    storage emulated)"""
    from oracle import gcc_oracle as O
    from tests.test_oracle_golden import build_sagan_oracle
    zz, real, vz, vreal = (torch.from_numpy(z['it0.' + k]) for k in ('z', 'real', 'vz', 'vreal'))
    O.EMULATE_BF16 = emulate
    try:
        om, ot, _ = build_sagan_oracle(z)
        for o in (om, ot):
            o.lr_G = o.lr_D = o.lr_arch = 0.0
        om.set_input(zz, real)
        om.optimize_parameters()
        g = {}
        for tag, who in (('t', ot), ('s', om)):
            for k in who.G_keys:
                g[(tag + 'G', k)] = who.G[k].grad.clone()
            for k in who.D_w_keys:
                g[(tag + 'D', k)] = who.D[k].grad.clone()
        for i in range(2):
            g[('T', i)] = om.T[i].grad.clone()
        om.set_input(vz, vreal)
        om.clipping_mask_alpha()
        om.optimizer_netD_arch()
        for k in om.D_a_keys:
            g[('alpha', k)] = om.D[k].grad.clone()
        return g
    finally:
        O.EMULATE_BF16 = False


def test_sagan_gradients_vs_oracle(golden_dir):
    """one iteration + arch step with every learning rate 0: every parameter gradient (u, v of the discriminators
    included) against the oracle's autograd gradient, fp32 and bf16-storage-emulated"""
    from oracle import gcc_oracle as O
    from tests.test_oracle_golden import build_sagan_oracle
    z = load(golden_dir, 'sagan_gcc.npz')
    model, teacher, opt = _build(z)
    for m in (model, teacher):
        for o in (m.optimizer_G, m.optimizer_D):
            o.param_groups[0]['lr'] = 0.0
    model.optimizer_arch.param_groups[0]['lr'] = 0.0
    zz, real, vz, vreal = (torch.from_numpy(z['it0.' + k]) for k in ('z', 'real', 'vz', 'vreal'))

    g32, g16 = _oracle_grads(z, False), _oracle_grads(z, True)
    model.set_input({'z': zz, 'real_img': real, 'img_path': ['p'] * 4})
    model.optimize_parameters()
    torch.cuda.synchronize()
    bad = []

    def check(key, g):
        g = g.float().cpu()
        zero = (key[0][1:] == 'G' and _ZERO_G(key[1])) or (key[0][1:] == 'D' and _ZERO_D(key[1]))
        if zero:
            return
        r32, r16, floor = _rel(g, g32[key]), _rel(g, g16[key]), _rel(g16[key], g32[key])
        print('%-6s %-30s vs fp32 %.4f  vs bf16-emulated %.4f  (emulated vs fp32 %.4f)' % (key[0], key[1], r32, r16, floor))
        # scalars (the attention gammas: sum dy o over the whole batch, a cancelling sum the emulation itself misses by
        # half): three noise floors, as for the small tensors of tests/test_srgan_gpu.py
        k = 3.0 if g.numel() == 1 else 1.5
        if not (r16 <= 6e-2 or r32 <= k * floor + 2e-2):
            bad.append((key, r32, r16, floor))
    for tag, net in (('tD', teacher.netD), ('tG', teacher.netG), ('sD', model.netD), ('sG', model.netG)):
        sd = net.state_dict(keep_vars=True)
        for (t, k) in g32:
            if t == tag:
                check((t, k), sd[k].grad)
    for i in range(2):
        check(('T', i), model.transform_convs[i].weight.grad)
    model.set_input({'z': vz, 'real_img': vreal, 'img_path': ['p'] * 4})
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()
    torch.cuda.synchronize()
    sd = model.netD.state_dict(keep_vars=True)
    for (t, k) in g32:
        if t == 'alpha':
            check((t, k), sd[k].grad)
    assert not bad, bad


FULL_SAGAN_ARGV = ['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '0', '--ngf', '48', '--ndf', '64',
                   '--teacher_ngf', '64', '--online_distillation', '--darts_discriminator', '--threshold', '0.1',
                   '--lambda_L1', '1', '--lambda_content', '1', '--lambda_gram', '1', '--arch_lr', '1e-4', '--batch_size', '64']


def test_sagan_full_width_iteration_vs_oracle():
    """BASELINE.json configs[3] at its real widths (student ngf 48 / masked D ndf 64, teacher ngf 64 / ndf 64, 64 x 64, batch
    64, z 128): one iteration + arch step of the HIP path against the oracle on the same recipe weights.  The generated
    images against the fp32 oracle (before any update acts); the loss scalars against the bf16-emulating oracle, the
    trajectory the HIP path must stay on behind the sign-like Adam steps (beta1 = 0), AND against the fp32 oracle (3e-2 before the
    first Adam step has acted, 6e-2 behind it); the report says which scalars only the emulation explains."""
    from collections import OrderedDict
    from gcc_amd.options import options
    from gcc_amd.models import get_model_class
    from oracle import gcc_oracle as O
    from tests.golden.recipe import recipe_state_dict, recipe_transform
    opt = options.parse(FULL_SAGAN_ARGV)
    opt.isTrain = True
    opt.teacher_ndf = 64
    cls = get_model_class(opt)
    model = cls(opt)
    topt = copy.deepcopy(opt)
    topt.ngf, topt.ndf = opt.teacher_ngf, opt.teacher_ndf
    topt.darts_discriminator = topt.online_distillation = False
    teacher = cls(topt)
    teacher.model_train()
    model.teacher_model = teacher
    model.init_distillation()
    teacher.init_distillation()
    sds = {}
    for tag, net, seed in (('sG', model.netG, 811), ('sD', model.netD, 812), ('tG', teacher.netG, 813), ('tD', teacher.netD, 814)):
        sds[tag] = recipe_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items()), seed)
        net.load_state_dict(sds[tag])
    Ts = [recipe_transform(t.weight.shape[0], t.weight.shape[1], 820 + i) for i, t in enumerate(model.transform_convs)]
    with torch.no_grad():
        for t, v in zip(model.transform_convs, Ts):
            t.weight.copy_(v.to(DEV))
    model.refresh_weights()
    teacher.refresh_weights()
    model.model_train()
    g = torch.Generator().manual_seed(93)
    N = 64
    zz, vz = torch.randn(N, opt.z_dim, generator=g), torch.randn(N, opt.z_dim, generator=g)
    real, vreal = torch.rand(N, 3, 64, 64, generator=g) * 2 - 1, torch.rand(N, 3, 64, 64, generator=g) * 2 - 1
    model.set_input({'z': zz, 'real_img': real, 'img_path': ['p'] * N})
    model.optimize_parameters()
    fake, tfake = model.fake_img.cpu(), teacher.fake_img.cpu()
    model.set_input({'z': vz, 'real_img': vreal, 'img_path': ['p'] * N})
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()
    got, tgot = model.get_current_losses(), teacher.get_current_losses()

    def run_oracle(emulate):
        O.EMULATE_BF16 = emulate
        try:
            oopt = O.Opt(ngf=48, ndf=64, teacher_ngf=64, teacher_ndf=64, gan_mode=opt.gan_mode, lr=opt.lr, lambda_L1=1.0,
                         lambda_content=1.0, lambda_gram=1.0)
            ot = O.SAGANOracle(oopt, copy.deepcopy(sds['tG']), copy.deepcopy(sds['tD']), masked=False)
            om = O.SAGANOracle(oopt, copy.deepcopy(sds['sG']), copy.deepcopy(sds['sD']), [t.clone() for t in Ts], masked=True,
                               teacher=ot)
            om.set_input(zz, real)
            om.optimize_parameters()
            imgs = (om.fake_img.detach().clone(), ot.fake_img.detach().clone())
            om.set_input(vz, vreal)
            om.clipping_mask_alpha()
            om.optimizer_netD_arch()
            return imgs, dict(om.losses), dict(ot.losses)
        finally:
            O.EMULATE_BF16 = False
    (ref_fake, ref_tfake), ref_l, ref_tl = run_oracle(False)
    (emu_fake, emu_tfake), emu_l, emu_tl = run_oracle(True)
    bad = []
    for what, a, r, e_ in (('fake_img', fake, ref_fake, emu_fake), ('teacher fake_img', tfake, ref_tfake, emu_tfake)):
        e, floor = (a - r).abs(), (e_ - r).abs()
        print('%s: max %.4g mean %.4g   (bf16-emulating oracle against fp32: max %.4g mean %.4g)' % (what, e.max(), e.mean(), floor.max(), floor.mean()))
        if not (e.max() <= max(2e-2, 1.5 * float(floor.max())) and e.mean() <= max(3e-3, 1.5 * float(floor.mean()))):
            bad.append((what, float(e.max()), float(e.mean())))
    assert len(set(emu_l) & set(got)) >= 8, (sorted(emu_l), sorted(got))
    from tests import _updates
    bars = _updates.LossBars('sagan-full-width', n_map=10 ** 9)
    for tag, gl, el, rl in (('S', got, emu_l, ref_l), ('T', tgot, emu_tl, ref_tl)):
        for k, v in el.items():
            if k not in gl:
                continue
            print('%s %-24s got %.5g  bf16-emulating oracle %.5g  fp32 oracle %.5g' % (tag, k, gl[k], v, rl[k]))
            bars.add('%s %s' % (tag, k), k, gl[k], rl[k], v)
            if not abs(gl[k] - v) <= 3e-2 * max(1.0, abs(v)):
                bad.append((tag, k, gl[k], v))
            # ... and against the fp32 oracle (this test judged against the emulating mode alone): 3e-2 for
            # the terms no Adam step has acted on yet, 6e-2 behind one (batch 64 averages the hinge flips that the 4-image fixture
            # shows as 10-20 %: measured worst 0.4 % of |ref|, profiles/r6a_test_report.txt)
            pre_step = k in ('D_real', 'D_fake', 'content', 'gram', 'L1')
            if not abs(gl[k] - rl[k]) <= (3e-2 if pre_step else 6e-2) * max(1.0, abs(rl[k])):
                bad.append((tag, k, gl[k], 'fp32 oracle', rl[k]))
    bars.check(max_emul_only=0.35, require=False)
    assert not bad, bad
