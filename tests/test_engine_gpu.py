"""Engine-level backward checks on shallow networks: MobileResnetEngine (2 residual blocks) and the InstanceNorm
PatchGAN -- forward tensors (relative L2 <= 2e-2), dL/dx and every parameter gradient against the oracle's autograd.
Gradient bar per tensor (relative L2): generator within 2e-2 of the oracle with bf16 storage emulated (measured < 1e-2;
InstanceNorm's backward subtracts two means, which amplifies activation rounding, so the same oracle in fp32 sits
5-16% away -- printed beside each tensor); discriminator within 5e-2 of the emulated oracle, or no farther from the
fp32 oracle than 1.5x the measured bf16-storage deviation + 1e-2."""
from collections import OrderedDict

import pytest
import torch

from tests.test_pix2pix_gpu import DEV, _rel, load_recipe

pytestmark = pytest.mark.gpu


def _to_nhwc(ops, x):
    buf = ops.new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], DEV)
    ops.nchw_to_nhwc(x.to(DEV).contiguous(), buf)
    return buf


def _rb(t):
    return t.bfloat16().float()


def test_mobile_resnet_engine_shallow_backward():
    from gcc_amd import engine, ops
    from gcc_amd.models.Pix2Pix import MobileResnetGenerator
    from oracle import gcc_oracle as O
    net = MobileResnetGenerator(ngf=16, n_blocks=2).to(DEV)
    load_recipe(net, 71)
    engine.FlatParams(list(net.parameters()), DEV)
    eng = engine.MobileResnetEngine(net, DEV)
    eng.hook_names = ['model.9', 'model.10', 'model.11']
    eng.repack()
    g = torch.Generator().manual_seed(3)
    N, H = 2, 32
    x = _rb(torch.rand(N, 3, H, H, generator=g) * 2 - 1)
    g_out = _rb(torch.randn(N, 3, H, H, generator=g) * 0.1)
    g_feat = [_rb(torch.randn(N, 64, H // 4, H // 4, generator=g) * 0.05) for _ in range(3)]
    # oracle (autograd on the same bf16-representable inputs): fp32, and with bf16 storage emulated
    def run_oracle(emulate):
        O.EMULATE_BF16 = emulate
        try:
            sd = OrderedDict((k, v.detach().float().cpu().contiguous().clone().requires_grad_(True))
                             for k, v in net.state_dict().items())
            xr = x.clone().requires_grad_(True)
            feats = OrderedDict()
            out_ref = O.mobile_resnet_forward(sd, xr, features=feats, hook_idx=(9, 10, 11))
            loss = (out_ref * g_out).sum() + sum((f * gf).sum() for f, gf in zip(feats.values(), g_feat))
            loss.backward()
            return sd, xr, feats, out_ref
        finally:
            O.EMULATE_BF16 = False
    sd, xr, feats, out_ref = run_oracle(False)
    sd16, xr16, _, _ = run_oracle(True)
    # HIP engine
    c = eng._ctx(N, H, H)
    ops.nhwc_copy(_to_nhwc(ops, x), 0, c.x_in, 0, 3)
    eng.forward(c)
    out = ops.nhwc_to_nchw(c.out, 3).cpu()
    print('image rel %.4f' % _rel(out, out_ref.detach()))
    assert _rel(out, out_ref.detach()) <= 2e-2
    for f, fr in zip(eng.features(c), feats.values()):
        assert _rel(f.float().cpu(), fr.detach()) <= 2e-2
    ops.nhwc_copy(_to_nhwc(ops, g_out), 0, c.g_out, 0, 3)
    dx = eng.backward(c, g_feat=[_to_nhwc(ops, t) for t in g_feat], need_dx=True)
    torch.cuda.synchronize()
    bad = []

    def check(name, got, ref32, ref16):
        r32, r16, floor = _rel(got, ref32), _rel(got, ref16), _rel(ref16, ref32)
        print('%-44s vs fp32 %.4f  vs bf16-emulated %.4f  (emulated vs fp32 %.4f)' % (name, r32, r16, floor))
        if not r16 <= 2e-2:          # measured <= 1e-2 against the emulating oracle on every tensor
            bad.append((name, r32, r16, floor))
    check('dL/dx', ops.nhwc_to_nchw(dx, 3).cpu(), xr.grad, xr16.grad)
    for k, p in net.state_dict(keep_vars=True).items():
        ref = sd[k].grad
        if k.endswith('.bias') and ref.abs().max() < 1e-4 * sd[k[:-4] + 'weight'].grad.abs().max():
            continue        # bias in front of an InstanceNorm: zero gradient
        check(k, p.grad.float().cpu(), ref, sd16[k].grad)
    assert not bad, bad


def test_instance_norm_patchgan_engine_backward():
    """CycleGAN's plain discriminator (InstanceNorm, biased convs): pred, dL/dx, parameter gradients, with an extra
    gradient injected at the two hooked features"""
    from gcc_amd import engine, ops
    from gcc_amd.models.CycleGAN import NLayerDiscriminator
    from oracle import gcc_oracle as O
    net = NLayerDiscriminator(input_nc=3, ndf=16).to(DEV)
    load_recipe(net, 72)
    engine.FlatParams(list(net.parameters()), DEV)
    eng = engine.PatchGANEngine(net, False, 0.5, DEV)
    eng.repack()
    g = torch.Generator().manual_seed(4)
    N, H = 2, 64
    x = _rb(torch.rand(N, 3, H, H, generator=g) * 2 - 1)
    g_pred = _rb(torch.randn(N, 1, 6, 6, generator=g) * 0.1)
    g_feat = [_rb(torch.randn(N, 32, 16, 16, generator=g) * 0.02), _rb(torch.randn(N, 128, 7, 7, generator=g) * 0.02)]

    def run_oracle(emulate):
        O.EMULATE_BF16 = emulate
        try:
            sd = OrderedDict((k, v.detach().float().cpu().contiguous().clone().requires_grad_(True))
                             for k, v in net.state_dict().items())
            xr = x.clone().requires_grad_(True)
            feats = OrderedDict()
            pred_ref = O.patchgan_forward(sd, xr, False, 0.5, True, features=feats, hook_names=['model.3', 'model.9'])
            ((pred_ref * g_pred).sum() + sum((f * gf).sum() for f, gf in zip(feats.values(), g_feat))).backward()
            return sd, xr, pred_ref
        finally:
            O.EMULATE_BF16 = False
    sd, xr, pred_ref = run_oracle(False)
    sd16, xr16, _ = run_oracle(True)
    c = eng.new_ctx(N, H, H, 't')
    ops.nhwc_copy(_to_nhwc(ops, x), 0, c.x_in, 0, 3)
    eng.forward(c)
    assert _rel(ops.nhwc_to_nchw(c.pred, 1).cpu(), pred_ref.detach()) <= 2e-2
    gp = eng.grad_pred_buffer(c)
    ops.nhwc_copy(_to_nhwc(ops, g_pred), 0, gp, 0, 1)
    dx = eng.backward(c, g_feat=[_to_nhwc(ops, t) for t in g_feat], wgrad=True, need_dx=True)
    torch.cuda.synchronize()
    bad = []

    def check(name, got, ref32, ref16):
        r32, r16, floor = _rel(got, ref32), _rel(got, ref16), _rel(ref16, ref32)
        print('%-20s vs fp32 %.4f  vs bf16-emulated %.4f  (emulated vs fp32 %.4f)' % (name, r32, r16, floor))
        if not (r16 <= 5e-2 or r32 <= 1.5 * floor + 1e-2):
            bad.append((name, r32, r16, floor))
    check('dL/dx', ops.nhwc_to_nchw(dx, 3).cpu(), xr.grad, xr16.grad)
    for k, p in net.state_dict(keep_vars=True).items():
        if k in ('model.2.bias', 'model.5.bias', 'model.8.bias'):
            continue
        check(k, p.grad.float().cpu(), sd[k].grad, sd16[k].grad)
    assert not bad, bad


@pytest.mark.parametrize('N,H,W', [(1, 32, 48), (3, 40, 24), (5, 16, 64)])
def test_engines_forward_non_square_odd_batch(N, H, W):
    """MobileResnet and SRResNet engines on geometries the golden fixtures do not cover (non-square maps, batch sizes
    that are not powers of two): forward against the oracle with bf16 storage emulated"""
    from gcc_amd import engine, ops
    from gcc_amd.models.Pix2Pix import MobileResnetGenerator
    from gcc_amd.models.SRGAN import Generator
    from oracle import gcc_oracle as O
    from tests.golden.recipe import srgan_condition
    g = torch.Generator().manual_seed(N + H + W)
    x = _rb(torch.rand(N, 3, H, W, generator=g) * 2 - 1)
    for kind in ('resnet', 'srresnet'):
        if kind == 'resnet':
            net = MobileResnetGenerator(ngf=16, n_blocks=3).to(DEV)
            load_recipe(net, 171)
        else:
            net = Generator(n_channels=16, n_blocks=2).to(DEV)
            load_recipe(net, 175)
            srgan_condition(net.state_dict())
        engine.FlatParams(list(net.parameters()), DEV)
        eng = engine.MobileResnetEngine(net, DEV) if kind == 'resnet' else engine.SRResNetEngine(net, DEV)
        eng.repack()
        sd = OrderedDict((k, v.detach().float().cpu().contiguous().clone()) for k, v in net.state_dict().items())
        O.EMULATE_BF16 = True
        try:
            ref = O.mobile_resnet_forward(sd, x) if kind == 'resnet' else O.srresnet_forward(sd, x, True)
        finally:
            O.EMULATE_BF16 = False
        c = eng._ctx(N, H, W)
        ops.nhwc_copy(_to_nhwc(ops, x), 0, c.x_in, 0, 3)
        if kind == 'resnet':
            eng.forward(c)
        else:
            eng.forward(c, train=True)
        out = ops.nhwc_to_nchw(c.out, 3).cpu()
        assert out.shape == ref.shape
        r = _rel(out, ref.detach())
        print('%s N%d %dx%d: rel to the bf16-emulating oracle %.2e' % (kind, N, H, W, r))
        assert r <= 1e-2, (kind, r)
