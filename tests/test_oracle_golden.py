"""Pins oracle/gcc_oracle.py against golden vectors produced by the real reference
(tests/golden/make_fixtures.py).  CPU only."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import gcc_oracle as O
from tests.golden.recipe import recipe_state_dict, recipe_transform, sample_idx

torch.set_num_threads(8)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def sd_from(z, prefix):
    sd = OrderedDict()
    for k in z.files:
        if k.startswith(prefix):
            v = torch.from_numpy(np.array(z[k]))
            sd[k[len(prefix):]] = v
    return sd


def test_eval_generated_images(golden_dir):
    z = load(golden_dir, 'pix2pix_eval_d8.npz')
    G = recipe_state_dict(O.unet_shapes(8, 8), int(z['seed_G']))
    A, B = torch.from_numpy(z['A']), torch.from_numpy(z['B'])
    real_A = A if str(z['direction']) == 'AtoB' else B
    with torch.no_grad():
        out = O.unet_forward(G, real_A, num_downs=8, train=False)
    ref = torch.from_numpy(z['fake_B'])
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-5


def test_state_dict_key_parity(golden_dir):
    z = load(golden_dir, 'ops.npz')
    assert list(O.unet_shapes(4, 6).keys()) == [str(k) for k in z['init.G_keys']]
    assert list(O.patchgan_shapes(4, 6, False).keys()) == [str(k) for k in z['init.D_keys']]
    g = load(golden_dir, 'pix2pix_gcc_d6.npz')
    masked_keys = [k[len('final.sD.'):] for k in g.files if k.startswith('final.sD.')]
    assert list(O.patchgan_shapes(8, 6, True).keys()) == masked_keys
    shp = O.unet_shapes(16, 6)
    for k in [k[len('final.tG.'):] for k in g.files if k.startswith('final.tG.')]:
        n = 1
        for d in shp[k]:
            n *= d
        assert g['final.tG.' + k].size == min(n, 2048), k


def test_gate_fwd_bwd(golden_dir):
    z = load(golden_dir, 'ops.npz')
    x = torch.from_numpy(z['gate.x']).requires_grad_(True)
    a = torch.from_numpy(z['gate.alpha']).requires_grad_(True)
    y = O.gate(x, a, 0.5)
    y.backward(torch.from_numpy(z['gate.dy']))
    assert np.array_equal(O.gate_mask(a.detach(), 0.5).numpy(), z['gate.mask'])
    np.testing.assert_allclose(y.detach().numpy(), z['gate.y'], rtol=0, atol=0)
    np.testing.assert_allclose(x.grad.numpy(), z['gate.dx'], rtol=0, atol=0)
    np.testing.assert_allclose(a.grad.numpy(), z['gate.dalpha'], rtol=1e-6, atol=1e-6)


def test_gan_losses_and_gram(golden_dir):
    z = load(golden_dir, 'ops.npz')
    pred = torch.from_numpy(z['gan.pred'])
    n = 0
    for k in z.files:
        if k.startswith('gan.') and k.count('.') == 3 and not k.endswith('.grad'):
            _, mode, real, ford = k.split('.')
            p = pred.clone().requires_grad_(True)
            l = O.gan_loss(mode, p, bool(int(real)), bool(int(ford)))
            l.backward()
            assert abs(float(l.detach()) - float(z[k])) < 1e-6, k
            np.testing.assert_allclose(p.grad.numpy(), z[k + '.grad'], atol=1e-7)
            n += 1
    assert n == 15
    np.testing.assert_allclose(O.gram(torch.from_numpy(z['gram.x'])).numpy(), z['gram.y'], atol=1e-6)


def test_lr_schedule(golden_dir):
    z = load(golden_dir, 'pix2pix_pretrain_d6.npz')
    ec, ne, nd, lr = z['sched']
    want = z['lr_after_epoch']
    got = [lr * O.lr_lambda_linear(e, int(ec), int(ne), int(nd)) for e in range(1, len(want) + 1)]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_init_statistics(golden_dir):
    z = load(golden_dir, 'ops.npz')
    sd = O.init_state_dict(O.unet_shapes(4, 6), torch.Generator().manual_seed(3))
    w = torch.cat([v.flatten() for k, v in sd.items() if v.dim() == 4])
    bw = torch.cat([v.flatten() for k, v in sd.items() if k.endswith('.weight') and v.dim() == 1])
    bb = torch.cat([v.flatten() for k, v in sd.items() if k.endswith('.bias') and v.dim() == 1 and
                    not k.startswith('model.model.3')])
    ref = z['init.stats']
    assert abs(float(w.mean()) - ref[0]) < 2e-3 and abs(float(w.std()) - ref[1]) < 2e-3
    assert abs(float(bw.mean()) - ref[2]) < 1e-2 and abs(float(bw.std()) - ref[3]) < 1e-2
    assert abs(float(bb.mean()) - ref[4]) < 0.3 and abs(float(bb.std()) - ref[5]) < 0.3
    assert float(sd['model.model.3.bias'].abs().max()) == 0.0


def _compare_sd(got, z, prefix, atol, rtol=1e-4, skip=None):
    worst = 0.0
    for k in z.files:
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        if skip is not None and skip(name.split('@')[0]):
            continue
        ref = z[k]
        g = got[name].detach().reshape(-1)
        g = g[sample_idx(g.numel())].numpy()
        if ref.dtype.kind in 'iu':
            assert int(g[0]) == int(ref.reshape(-1)[0]), name
            continue
        err = np.abs(g - ref).max()
        tol = atol + rtol * np.abs(ref).max()
        assert err <= tol, (name, err, tol)
        worst = max(worst, err)
    return worst


def test_pretrain_two_iterations(golden_dir):
    z = load(golden_dir, 'pix2pix_pretrain_d6.npz')
    opt = O.Opt(ngf=8, ndf=8, num_downs=6, no_dropout=True, lambda_scale=1e-2, darts_discriminator=False,
                online_distillation=False, direction=str(z['direction']))
    sG, sD = [int(v) for v in z['seeds']]
    m = O.Pix2PixOracle(opt, recipe_state_dict(O.unet_shapes(8, 6), sG),
                        recipe_state_dict(O.patchgan_shapes(8, 6, False), sD), masked=False)
    for it in range(2):
        m.set_input(torch.from_numpy(z['it%d.A' % it]), torch.from_numpy(z['it%d.B' % it]))
        m.optimize_parameters()
        for k in ('G_GAN', 'G_L1', 'D_real', 'D_fake'):
            assert abs(m.losses[k] - float(z['it%d.loss.%s' % (it, k)])) < 1e-4 * max(1, abs(m.losses[k])), (it, k)
    _compare_sd(m.G, z, 'final.G.', atol=2e-5)
    _compare_sd(m.D, z, 'final.D.', atol=2e-5)


def build_gcc_oracle(z):
    """student+teacher oracle with the recipe weights the fixture was generated from"""
    opt = O.Opt(ngf=8, ndf=8, teacher_ngf=16, teacher_ndf=16, num_downs=6, no_dropout=True,
                direction=str(z['direction']), threshold=float(z['threshold']))
    s_sG, s_sD, s_tG, s_tD, s_T = [int(v) for v in z['seeds']]
    teacher = O.Pix2PixOracle(opt, recipe_state_dict(O.unet_shapes(16, 6), s_tG),
                              recipe_state_dict(O.patchgan_shapes(16, 6, False), s_tD), masked=False)
    sD = recipe_state_dict(O.patchgan_shapes(8, 6, True), s_sD)
    for k in z.files:
        if k.startswith('init.sD.'):
            sD[k[len('init.sD.'):]] = torch.from_numpy(np.array(z[k]))
    sw, tw = [16, 64, 128, 32], [32, 128, 256, 64]
    T = [recipe_transform(t, s, s_T + i) for i, (s, t) in enumerate(zip(sw, tw))]
    m = O.Pix2PixOracle(opt, recipe_state_dict(O.unet_shapes(8, 6), s_sG), sD, T, masked=True, teacher=teacher)
    return m, teacher, opt


def test_gcc_two_iterations(golden_dir):
    z = load(golden_dir, 'pix2pix_gcc_d6.npz')
    m, teacher, opt = build_gcc_oracle(z)
    for it in range(2):
        m.set_input(torch.from_numpy(z['it%d.A' % it]), torch.from_numpy(z['it%d.B' % it]))
        m.optimize_parameters()
        if it == 0:
            np.testing.assert_allclose(m.fake_B.numpy(), z['it0.fake_B'], atol=2e-5)
            np.testing.assert_allclose(teacher.fake_B.numpy(), z['it0.Tfake_B'], atol=2e-5)
            for j in range(6):
                ref = z['it0.target.%d' % j]
                np.testing.assert_allclose(m.targets[j].numpy(), ref, atol=2e-5 + 1e-4 * np.abs(ref).max())
            for j in range(4):
                ref = z['it0.sfeat.%d' % j]
                np.testing.assert_allclose(list(m.g_feats.values())[j].detach().numpy(), ref,
                                           atol=2e-5 + 1e-4 * np.abs(ref).max())
            for j in range(2):
                ref = z['it0.tDfeat_on_sfake.%d' % j]
                np.testing.assert_allclose(m.dist_feats[4 + j].numpy(), ref, atol=2e-5 + 1e-4 * np.abs(ref).max())
            _compare_sd(m.G, z, 'it0.afterstep.sG.', atol=2e-5)
            _compare_sd(m.D, z, 'it0.afterstep.sD.', atol=2e-5)
        m.set_input(torch.from_numpy(z['it%d.vA' % it]), torch.from_numpy(z['it%d.vB' % it]))
        m.clipping_mask_alpha()
        m.optimizer_netD_arch()
        for k in z.files:
            if k.startswith('it%d.loss.' % it):
                name = k.split('.')[-1]
                ref = float(z[k])
                assert abs(m.losses[name] - ref) <= 2e-4 * max(1.0, abs(ref)), (it, name, m.losses[name], ref)
            if k.startswith('it%d.tloss.' % it):
                name = k.split('.')[-1]
                ref = float(z[k])
                assert abs(teacher.losses[name] - ref) <= 2e-4 * max(1.0, abs(ref)), (it, name)
    _compare_sd(m.G, z, 'final.sG.', atol=3e-5)
    _compare_sd(m.D, z, 'final.sD.', atol=3e-5)
    _compare_sd(teacher.G, z, 'final.tG.', atol=3e-5)
    _compare_sd(teacher.D, z, 'final.tD.', atol=3e-5)
    for i in range(4):
        np.testing.assert_allclose(m.T[i].detach().numpy(), z['final.T.%d' % i], atol=3e-5)


def test_prune_cfgs_bit_exact(golden_dir):
    z = load(golden_dir, 'prune_d8.npz')
    G = recipe_state_dict(O.unet_shapes(8, 8), int(z['seed_G']))
    gsp = torch.Generator().manual_seed(302)
    # same BN-scale spread the fixture script applied, in the reference's modules() order
    for name in O._unet_bn_names(8):
        G[name + '.weight'] = 1.0 + 0.02 * torch.randn(G[name + '.weight'].shape, generator=gsp)
    mx, mn = O.max_min_bn_scale(G)
    assert [mx, mn] == [float(v) for v in z['bn.max_min']]
    for i, t in enumerate(z['bn.thresholds']):
        f, c = O.scale_prune_cfg(G, float(t), ngf=8)
        assert f == [int(v) for v in z['bn.f.%d' % i]], (i, t)
        assert c == [int(v) for v in z['bn.c.%d' % i]], (i, t)
    mx, mn = O.max_min_conv_norm(G)
    assert [mx, mn] == [float(v) for v in z['norm.max_min']]
    for i, t in enumerate(z['norm.thresholds']):
        f, c = O.norm_prune_cfg(G, float(t), ngf=8)
        assert f == [int(v) for v in z['norm.f.%d' % i]], (i, t)
        assert c == [int(v) for v in z['norm.c.%d' % i]], (i, t)


def test_pruned_student_irregular_widths(golden_dir):
    """generator built from filter_cfgs / channel_cfgs (widths such as 6, 21, 31, 29): eval image and one
    training iteration against the reference"""
    z = load(golden_dir, 'pix2pix_pruned_d8.npz')
    f, c = [int(v) for v in z['f']], [int(v) for v in z['c']]
    sG, sD = [int(v) for v in z['seeds']]
    G = recipe_state_dict(O.unet_shapes_cfg(f, c), sG)
    D = recipe_state_dict(O.patchgan_shapes(8, 6, False), sD)
    A, B = torch.from_numpy(z['A']), torch.from_numpy(z['B'])
    real_A = A if str(z['direction']) == 'AtoB' else B
    with torch.no_grad():
        out = O.unet_forward(G, real_A, 8, train=False)
    np.testing.assert_allclose(out.numpy(), z['eval.fake_B'], atol=2e-5)
    opt = O.Opt(ngf=8, ndf=8, num_downs=8, no_dropout=True, darts_discriminator=False, online_distillation=False,
                direction=str(z['direction']))
    m = O.Pix2PixOracle(opt, G, D, masked=False)
    m.set_input(A, B)
    m.optimize_parameters()
    np.testing.assert_allclose(m.fake_B.numpy(), z['train.fake_B'], atol=2e-5)
    for k in ('G_GAN', 'G_L1', 'D_real', 'D_fake'):
        assert abs(m.losses[k] - float(z['loss.' + k])) < 1e-4 * max(1, abs(m.losses[k]))
    _compare_sd(m.G, z, 'final.G.', atol=2e-5)


def build_resnet_gcc_oracle(z):
    """--backbone resnet student (ngf 8) + teacher (ngf 16) with the recipe weights of pix2pix_resnet_gcc.npz"""
    opt = O.Opt(ngf=8, ndf=8, teacher_ngf=16, teacher_ndf=16, direction=str(z['direction']), backbone='resnet')
    s_sG, s_sD, s_tG, s_tD, s_T = [int(v) for v in z['seeds']]
    teacher = O.Pix2PixOracle(opt, recipe_state_dict(O.mobile_resnet_shapes(16), s_tG),
                              recipe_state_dict(O.patchgan_shapes(16, 6, False), s_tD), masked=False)
    sD = recipe_state_dict(O.patchgan_shapes(8, 6, True), s_sD)
    sD['model.2.alpha'][0] = 0.3
    T = [recipe_transform(64, 32, s_T + i) for i in range(4)]
    m = O.Pix2PixOracle(opt, recipe_state_dict(O.mobile_resnet_shapes(8), s_sG), sD, T, masked=True, teacher=teacher)
    return m, teacher, opt


def test_resnet_backbone_gcc_iteration(golden_dir):
    """MobileResnetGenerator (separable convs + InstanceNorm + reflect padding) under the Pix2Pix GCC step"""
    z = load(golden_dir, 'pix2pix_resnet_gcc.npz')
    assert list(O.mobile_resnet_shapes(8).keys()) == [str(k) for k in z['G_keys']]
    m, teacher, opt = build_resnet_gcc_oracle(z)
    A, B = torch.from_numpy(z['A']), torch.from_numpy(z['B'])
    real_A = A if opt.direction == 'AtoB' else B
    with torch.no_grad():
        out = O.mobile_resnet_forward(m.G, real_A)
    np.testing.assert_allclose(out.numpy(), z['eval.fake_B'], atol=2e-5)
    m.set_input(A, B)
    m.optimize_parameters()
    np.testing.assert_allclose(m.fake_B.numpy(), z['train.fake_B'], atol=2e-5)
    np.testing.assert_allclose(teacher.fake_B.numpy(), z['train.Tfake_B'], atol=2e-5)
    for j in range(4):
        ref = z['sfeat.%d' % j]
        np.testing.assert_allclose(list(m.g_feats.values())[j].detach().numpy(), ref, atol=2e-5 + 1e-4 * np.abs(ref).max())
    for j in range(6):
        ref = z['target.%d' % j]
        np.testing.assert_allclose(m.targets[j].numpy(), ref, atol=2e-5 + 1e-4 * np.abs(ref).max())
    m.set_input(torch.from_numpy(z['vA']), torch.from_numpy(z['vB']))
    m.clipping_mask_alpha()
    m.optimizer_netD_arch()
    for k in z.files:
        if k.startswith('loss.'):
            ref = float(z[k])
            assert abs(m.losses[k[5:]] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, m.losses[k[5:]], ref)
    # biases in front of an InstanceNorm have an analytically zero gradient (rounding noise through Adam): skip them
    skip = lambda k: k.endswith('.bias') and not k.startswith('model.26')
    _compare_sd(m.G, z, 'final.sG.', atol=3e-5, skip=skip)
    _compare_sd(teacher.G, z, 'final.tG.', atol=3e-5, skip=skip)
    _compare_sd(m.D, z, 'final.sD.', atol=3e-5)
