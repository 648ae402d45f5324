"""Kernel-level parity: every libgcc_hip.so entry point against a plain PyTorch-CPU fp32 reference
of the same op on the same (bf16-rounded) inputs.  Tolerance: outputs are bf16 (8 significant
bits) accumulated in fp32 -> |err| <= 1.2e-2 * max|ref| (+ tiny absolute floor); integer/mask
outputs exact."""
import math
import os

import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _ops():
    from gcc_amd import ops
    return ops


def rb(x):
    """round to bf16 and back (CPU)"""
    return x.bfloat16().float()


def to_dev(x, ld=None):
    ops = _ops()
    N, C, H, W = x.shape
    t = ops.new_act(N, C, H, W, DEV, ld=ld)
    t.copy_(x.bfloat16().to(DEV))
    return t


def to_cpu(t):
    return t.float().cpu()


def close(got, ref, tol=1.2e-2, floor=1e-6, what=''):
    err = (got - ref).abs().max().item()
    lim = tol * ref.abs().max().item() + floor
    assert err <= lim, '%s: err %.4g > %.4g (max|ref| %.4g)' % (what, err, lim, ref.abs().max().item())


def master_cl(w):
    """fp32 [Co,Ci,k,k] -> channels_last device parameter"""
    return w.to(DEV).contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # N, H, W, Ci, Co, k, s, p
    (2, 16, 16, 64, 128, 4, 2, 1),
    (2, 16, 16, 6, 128, 4, 2, 1),      # first PatchGAN layer: 6 channels in an 8-wide buffer
    (1, 9, 9, 128, 256, 4, 1, 1),      # stride-1 k4 (PatchGAN L4), odd size
    (2, 8, 8, 64, 1, 4, 1, 1),         # single output channel (PatchGAN L5)
    (2, 16, 16, 32, 3, 4, 2, 1),       # 3 output channels
    (2, 12, 12, 64, 96, 1, 1, 0),      # 1x1 transform conv
    (1, 10, 10, 24, 40, 3, 1, 1),      # channels not a multiple of 64 / 16
    (3, 2, 2, 256, 256, 4, 2, 1),      # U-Net bottleneck 2x2 -> 1x1
    (1, 32, 32, 3, 32, 4, 2, 1),       # U-Net first layer
    (2, 4, 4, 512, 512, 4, 2, 1),      # U-Net bottleneck, K = 8192: split-K path (fprop and dgrad)
    (4, 12, 12, 1024, 1, 4, 1, 1),     # PatchGAN head, Cout = 1, K = 16384: split-K path
    (16, 2, 2, 256, 384, 4, 2, 1),     # split-K with an N tail (384 = 3 x 128)
    (2, 32, 32, 128, 128, 4, 2, 1),    # 256x128 tile candidates (the tile256x128 / tile256x256 plans below force them)
    (2, 16, 16, 256, 384, 4, 1, 1),    # 256-pixel tiles with M and N tails, fprop and dgrad
    (3, 20, 20, 128, 512, 4, 2, 1),    # 256x256 tiles, ragged M
    (2, 22, 22, 3, 16, 7, 1, 0),       # MobileResnet stem: 7x7 on the reflect-padded image
    (2, 22, 22, 16, 3, 7, 1, 0),       # MobileResnet head: 7x7 to 3 channels
    (2, 16, 16, 16, 32, 3, 2, 1),      # MobileResnet down conv k3 s2; its dgrad is ConvTranspose(k3,s2,p1,output_padding=1)
    (1, 16, 16, 128, 64, 3, 2, 1),
    (3, 37, 29, 6, 72, 4, 2, 1),       # thin-input kernel: ragged pixel tiles, 72 = 2 x 32 + 8 channels
    (2, 18, 18, 3, 40, 3, 1, 1),       # thin-input kernel: 9 taps (padded to 12), 40 channels
    (1, 64, 64, 5, 200, 4, 2, 1),      # thin-input kernel: two 128-channel columns
    (3, 20, 28, 5, 56, 4, 2, 1),       # thin-output data gradient: ragged pair tiles (Wo = 14), 56 channels in
    (2, 36, 36, 3, 40, 4, 2, 1),       # thin-output data gradient: two pair tiles per row, 40 channels in
    (4, 33, 33, 264, 520, 4, 1, 1),    # wgrad 256x256 tiles: ragged columns (4224) and output channels (520), 4 pixel splits
    (4, 32, 32, 64, 64, 4, 2, 1),      # 128 x 64 tiles on the uniform-tap path: the three-stage k loop (fprop and dgrad), 16 / 4 k-steps
    (4, 32, 32, 128, 32, 4, 2, 1),     # 128 x 32 tiles, three stages, 32 k-steps
    (3, 20, 36, 64, 48, 4, 2, 1),      # ... with ragged pixel tiles and a channel tail
    (2, 8, 8, 64, 64, 1, 1, 0),        # a single k-step under the three-stage loop (1x1)
]


# fprop / dgrad tile plans (include/gcc_hip.h gcc_conv_t.plan: they travel with the call): every geometry runs on every tile
# family it can be routed to -- the default plan picks the 256-pixel tiles only for chip-filling grids, which small test cases
# never are
PLANS = {'default': {}, 'tile128': dict(tile_families=1), 'tile256x128': dict(tile_families=2, big_min=1, big_nk=1),
         'tile256x256': dict(tile_families=3, big_min=1, big_nk=1),
         'tile256x256_pair': dict(tile_families=3, big_min=1, big_nk=1, pair=1)}   # two workgroups per tile, K halves combined inside the launch


def _tiles(case):
    import ctypes as C
    from gcc_amd import _lib
    from gcc_amd import ops
    N, H, W, Ci, Co, k, s, p = case
    d = ops.conv_desc(N, H, W, Ci, Co, k, s, p, (Ci + 7) & ~7, (Co + 7) & ~7)         # under the calling thread's current plan
    return (_lib.load().gcc_conv_tile(C.byref(d), 0), _lib.load().gcc_conv_tile(C.byref(d), 1))


@pytest.fixture
def conv_plan(request):
    from gcc_amd import ops
    ops.set_plan()

    def choose(name):
        ops.set_plan(**PLANS[name])
    yield choose
    ops.set_plan()


@pytest.mark.parametrize('plan', list(PLANS))
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fprop_dgrad_wgrad(case, plan, conv_plan):
    ops = _ops()
    N, H, W, Ci, Co, k, s, p = case
    base_tiles = _tiles(case)
    conv_plan(plan)
    tiles = _tiles(case)
    if plan != 'default' and tiles == base_tiles:
        pytest.skip('plan %s routes this geometry to the default tiles %s' % (plan, tiles))
    if plan.endswith('_pair') and 256256 not in tiles:
        pytest.skip('no 256x256 tiles for this geometry: nothing to pair')
    print('plan %s: fprop tile %d, dgrad tile %d' % (plan, tiles[0], tiles[1]))
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, k, k, generator=g) * 0.1)
    b = torch.randn(Co, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    xd = to_dev(x)
    m = master_cl(w)
    wp, wtp = ops.pack_weights(m)
    # packing check
    close(wp.float().cpu()[:, :, :Ci], w.permute(0, 2, 3, 1).reshape(Co, k * k, Ci), tol=0, floor=0, what='pack W')
    close(wtp.float().cpu()[:, :, :Co], w.permute(1, 2, 3, 0).reshape(Ci, k * k, Co), tol=0, floor=0, what='pack Wt')
    # forward, with stats
    y, stats = ops.conv_fprop(xd, wp, Co, k, s, p, want_stats=True)
    torch.cuda.synchronize()
    yg = to_cpu(y)
    close(yg, y_ref.detach(), what='fprop')
    st = stats.sum(0).cpu()
    close(st[0], yg.sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='stats sum')
    close(st[1], (yg * yg).sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='stats sumsq')
    # forward with bias + leaky relu epilogue
    y2 = ops.conv_fprop(xd, wp, Co, k, s, p, bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2)
    close(to_cpu(y2), F.leaky_relu(y_ref.detach() + b[None, :, None, None], 0.2), what='fprop+bias+lrelu')
    # pad channels of the output buffer stay zero
    base = y2.permute(0, 2, 3, 1)
    ld = y2.stride(3)
    if ld > Co:
        full = torch.as_strided(y2, (N, ld, y2.shape[2], y2.shape[3]), y2.stride())
        assert float(full[:, Co:].float().abs().max()) == 0.0
    # backward data
    dyd = to_dev(dy)
    dx = ops.conv_dgrad(dyd, wtp, Ci, H, W, k, s, p)
    close(to_cpu(dx), xr.grad, what='dgrad')
    # backward weight (fresh and accumulating)
    dw = torch.zeros_like(m)
    ops.conv_wgrad(xd, dyd, dw, k, s, p, accumulate=False)
    close(dw.cpu(), wr.grad, tol=5e-3, floor=1e-4, what='wgrad')
    ops.conv_wgrad(xd, dyd, dw, k, s, p, accumulate=True)
    close(dw.cpu(), 2 * wr.grad, tol=5e-3, floor=1e-4, what='wgrad accumulate')


# wgrad_ts_kernel (conv_wgrad.hip): the tap-stationary k4 s1 p1 / k3 s1 p1 weight gradient.  GCC_OPT_WGRAD_TS = 2 routes every fitting
# geometry to it; WGS_BIG sets the number of pixel splits (1 = direct write / accumulate into dW, more = slabs + fold).
TS_CASES = [
    # k, N, H, W, Ci, Co, workgroups aimed at
    (4, 2, 9, 21, 64, 64, 1),          # one channel tile, ragged 8 x 16 blocks (Ho = 8, Wo = 20), no split: direct write
    (4, 2, 9, 21, 64, 64, 4),          # ... 4 blocks over 4 splits: a single block per workgroup
    (4, 3, 32, 32, 128, 64, 2),        # 31 x 31 outputs on the 32 x 32 block grid: 24 blocks, one split
    (4, 3, 32, 32, 128, 64, 12),       # ... 6 splits of 4 blocks (the three-stage loop with its tail)
    (4, 1, 5, 40, 64, 192, 9),         # short image (one block row of 4 lines), 3 blocks per image
    (4, 2, 17, 17, 192, 128, 6),       # Ho = Wo = 16: exactly one block wide, two deep
    (4, 16, 32, 32, 512, 1024, 256),   # the discriminators' L4 at the headline batch (the layer the kernel exists for), 2 splits
    # k3 s1 p1 (round 6): nine waves, one tap each
    (3, 2, 9, 21, 64, 64, 1),          # ragged blocks (9 x 21 outputs on 8 x 16 blocks), no split: direct write
    (3, 2, 9, 21, 64, 64, 8),          # ... 8 blocks over 8 splits: a single block per workgroup
    (3, 3, 32, 32, 128, 64, 2),        # 24 blocks, one split per tile
    (3, 3, 32, 32, 128, 64, 12),       # ... 6 splits of 4 blocks
    (3, 1, 5, 40, 64, 192, 9),         # short image, three channel tiles
    (3, 2, 16, 16, 192, 128, 6),       # exactly one block wide, two deep
    (3, 16, 96, 96, 64, 64, 144),      # SRGAN's trunk layer at the 96 -> 384 size (the layer this form exists for): 144 splits of 8 blocks
    (3, 4, 192, 192, 64, 128, 64),     # a stride-1 layer of its discriminator (N = 4 of 16)
]


@pytest.mark.parametrize('case', TS_CASES)
def test_wgrad_tap_stationary(case):
    ops = _ops()
    from gcc_amd import _lib
    lib = _lib.load()
    k, N, H, W, Ci, Co, wgs = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    Ho, Wo = (H - 1, W - 1) if k == 4 else (H, W)
    dy = rb(torch.randn(N, Co, Ho, Wo, generator=g))
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(x, w, None, stride=1, padding=1).backward(dy)
    xd, dyd = to_dev(x), to_dev(dy)
    m = master_cl(w.detach())
    try:
        ops.set_plan(wgrad_wgs_big=wgs)
        lib.gcc_set_option(_lib.OPT_WGRAD_TS, 2)
        dw = torch.full_like(m, 7.0)                 # stale contents must not survive a fresh gradient
        ops.lib().gcc_launch_count(1)
        ops.conv_wgrad(xd, dyd, dw, k, 1, 1, accumulate=False)
        launches = int(ops.lib().gcc_launch_count(1))
        ops.conv_wgrad(xd, dyd, dw, k, 1, 1, accumulate=True)
        torch.cuda.synchronize()
        lib.gcc_set_option(_lib.OPT_WGRAD_TS, 0)
        dw0 = torch.zeros_like(m)
        ops.conv_wgrad(xd, dyd, dw0, k, 1, 1, accumulate=False)
        torch.cuda.synchronize()
    finally:
        lib.gcc_set_option(_lib.OPT_WGRAD_TS, -1)
        ops.set_plan()
    tiles = (Ci // 64) * (Co // 64)
    assert launches == (1 if wgs <= tiles else 2), launches      # no split: the kernel alone
    scale = float(w.grad.abs().max())
    close(dw.cpu(), 2 * w.grad, tol=5e-3, floor=1e-4 * max(scale, 1.0), what='tap-stationary wgrad (fresh + accumulate)')
    # against wgrad_kernel on the same operands: fp32 sums of the same bf16 products in another order
    close(dw.cpu(), 2 * dw0.cpu(), tol=1e-4, floor=1e-5 * max(scale, 1.0), what='tap-stationary vs column-tiled wgrad')


# True shapes of the headline configuration (BASELINE.json configs[1]: N = 16 per GPU, ndf 128, teacher ngf 64, 256 x 256) under
# the DEFAULT plan: what bench.py actually launches (igemm 256x256 / 256x128 tiles, the 256x256 weight-gradient tiles).
TRUE_SHAPES = {
    # name: (N, H, W, Ci, Co, k, s, p)          reference layer
    'patchgan_L2': (16, 128, 128, 128, 256, 4, 2, 1),     # models/Pix2Pix.py:287-300, n = 1
    'patchgan_L3': (16, 64, 64, 256, 512, 4, 2, 1),       # n = 2
    'patchgan_L4': (16, 32, 32, 512, 1024, 4, 1, 1),      # stride-1 layer -> 31 x 31
    'teacherG_d2': (16, 64, 64, 128, 256, 4, 2, 1),       # U-Net ngf 64 down conv at depth 2 (models/Pix2Pix.py:31-32)
    'teacherG_u2': (16, 64, 64, 128, 512, 4, 2, 1),       # ConvTranspose2d(512 -> 128) of depth 2 as its adjoint conv (:40-56)
}


@pytest.mark.parametrize('pair', [0, 1])
@pytest.mark.parametrize('name', list(TRUE_SHAPES))
def test_conv_true_shapes_default_plan(name, pair):
    """pair = 1: the single-stream plan (what bench.py's bracketed roofline step and --serialize-streams run): half-chip 256x256
    launches (PatchGAN L3 forward, L4 data gradient) split K over two workgroups per tile"""
    ops = _ops()
    from gcc_amd import _lib
    if pair and name not in ('patchgan_L3', 'patchgan_L4'):
        pytest.skip('no half-chip 256x256 launch in this layer')
    ops.set_plan(pair=pair)
    try:
        _true_shape_case(ops, name)
    finally:
        ops.set_plan()


def _true_shape_case(ops, name):
    case = TRUE_SHAPES[name]
    N, H, W, Ci, Co, k, s, p = case
    tiles = _tiles(case)
    print('%s: fprop tile %d, dgrad tile %d' % (name, tiles[0], tiles[1]))
    if name.startswith('patchgan'):
        # forward: 256-pixel tiles; data gradient: L3 / L4 too (L2's stride-2 phases have K = 4 taps x 256 = 16 k-steps, below the
        # plan's depth threshold: 128-pixel tiles)
        assert tiles[0] // 1000 == 256, 'the PatchGAN layers run on the 256-pixel tiles'
        assert tiles[1] // 1000 == (128 if name == 'patchgan_L2' else 256)
    g = torch.Generator().manual_seed(len(name))
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, k, k, generator=g) * 0.02)
    with torch.no_grad():
        y_ref = F.conv2d(x, w, None, stride=s, padding=p)
        dy = rb(torch.randn(y_ref.shape, generator=g))
        dx_ref = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=s, padding=p)
        dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=s, padding=p)
    xd = to_dev(x)
    m = master_cl(w)
    wp, wtp = ops.pack_weights(m)
    y, stats = ops.conv_fprop(xd, wp, Co, k, s, p, want_stats=True)
    yg = to_cpu(y)
    close(yg, y_ref, what=name + ' fprop')
    st = stats.sum(0).cpu()
    close(st[0], yg.sum((0, 2, 3)), tol=1e-3, floor=1e-2, what=name + ' stats sum')
    close(st[1], (yg * yg).sum((0, 2, 3)), tol=1e-3, floor=1e-2, what=name + ' stats sumsq')
    dyd = to_dev(dy)
    dx = ops.conv_dgrad(dyd, wtp, Ci, H, W, k, s, p)
    close(to_cpu(dx), dx_ref, what=name + ' dgrad')
    dw = torch.zeros_like(m)
    ops.conv_wgrad(xd, dyd, dw, k, s, p, accumulate=False)
    close(dw.cpu(), dw_ref, tol=5e-3, floor=1e-4, what=name + ' wgrad')


@pytest.mark.parametrize('case', [
    # N, H, W, Ci, Co, transposed, drop      (k4 s2 p1: the U-Net layers)
    (16, 8, 8, 256, 256, False, 0.0),        # student d5: 4x4 out, K = 4096: split K -> the fused fold kernel
    (16, 4, 4, 512, 512, False, 0.0),        # teacher d6
    (16, 2, 2, 256, 512, True, 0.5),         # student u6 (ConvT 512 -> 256, 2x2 -> 4x4) with dropout
    (16, 8, 8, 256, 512, True, 0.5),         # u4-like: 4 phases x 256 rows
    (3, 6, 10, 40, 72, False, 0.0),          # ragged sizes, channels not multiples of 64
    (16, 32, 32, 128, 256, False, 0.0),      # un-split layer: the ordinary three kernels inside the one call
    (2, 16, 16, 64, 128, True, 0.0),
])
@pytest.mark.parametrize('fuse', [3, 2, 1, 0])
def test_conv_bn_act_one_call(case, fuse):
    """gcc_conv_bn_act against conv -> BatchNorm2d(train) -> activation in fp32 torch on the bf16-rounded conv output: raw output,
    batch statistics, running statistics, both activated copies; dropout: the mask is the one gcc_bnact_bwd regenerates
    (checked through the backward of the same layer), rate ~ 0.5"""
    ops = _ops()
    from gcc_amd import _lib
    import torch.nn as nn
    N, H, W, Ci, Co, transposed, drop = case
    g = torch.Generator().manual_seed(N * 100 + H + Ci)
    lib = _lib.load()
    lib.gcc_set_option(_lib.OPT_FUSE_BN, fuse)
    try:
        bn = nn.BatchNorm2d(Co if not transposed else Ci).to(DEV)
        Cn = bn.num_features
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(Cn, generator=g))
            bn.bias.copy_(torch.randn(Cn, generator=g))
        rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
        if not transposed:
            x = rb(torch.randn(N, Ci, H, W, generator=g))
            wgt = rb(torch.randn(Co, Ci, 4, 4, generator=g) * 0.05)
            ref_raw = F.conv2d(x, wgt, None, stride=2, padding=1)
            wp, _ = ops.pack_weights(master_cl(wgt))
            src, wpk = to_dev(x), wp
        else:                                   # ConvTranspose2d(Co -> Ci): adjoint conv Ci -> Co, forward = dgrad
            x = rb(torch.randn(N, Co, H // 2, W // 2, generator=g))
            wgt = rb(torch.randn(Co, Ci, 4, 4, generator=g) * 0.05)
            ref_raw = F.conv_transpose2d(x, wgt, None, stride=2, padding=1)
            _, wtp = ops.pack_weights(master_cl(wgt))
            src, wpk = to_dev(x), wtp
        raw = ops.new_act(*ref_raw.shape, DEV)
        y = ops.new_act(*ref_raw.shape, DEV)
        y2 = ops.new_act(*ref_raw.shape, DEV)
        st = ops.BNState(Cn, DEV)
        count = ref_raw.shape[0] * ref_raw.shape[2] * ref_raw.shape[3]
        seed = 1234567
        ops.conv_bn_act(transposed, src, wpk, raw, 4, 2, 1, bn, st, count, y, y2, act=ops.ACT_LRELU if not transposed else ops.ACT_RELU,
                        act2=ops.ACT_RELU, drop_p=drop, seed=seed)
        torch.cuda.synchronize()
        rawg = to_cpu(raw)
        close(rawg, ref_raw, what='raw conv output')
        mean = rawg.mean((0, 2, 3))
        var = rawg.var((0, 2, 3), unbiased=False)
        close(st.mean.cpu(), mean, tol=1e-4, floor=1e-5, what='batch mean')
        close(st.rstd.cpu(), 1.0 / torch.sqrt(var + bn.eps), tol=1e-4, floor=1e-5, what='rstd')
        close(bn.running_mean.cpu(), 0.9 * rm0.cpu() + 0.1 * mean, tol=1e-4, floor=1e-5, what='running mean')
        close(bn.running_var.cpu(), 0.9 * rv0.cpu() + 0.1 * var * count / (count - 1), tol=1e-4, floor=1e-5, what='running var')
        z = (rawg - mean[None, :, None, None]) / torch.sqrt(var + bn.eps)[None, :, None, None] * bn.weight.detach().cpu()[None, :, None, None] \
            + bn.bias.detach().cpu()[None, :, None, None]
        yg, y2g = to_cpu(y), to_cpu(y2)
        act1 = (lambda t: F.leaky_relu(t, 0.2)) if not transposed else torch.relu
        if drop == 0.0:
            close(yg, act1(z), what='activated output')
            close(y2g, torch.relu(z), what='second activated output')
        else:
            # the mask the backward pass regenerates from (seed, pixel, channel): a bare dropout backward of ones
            gup = to_dev(torch.ones(ref_raw.shape))
            dxb = ops.new_act(*ref_raw.shape, DEV)
            ops.bnact_bwd(raw, None, gup, dxb, bn=None, act=ops.ACT_NONE, drop_p=drop, seed=seed)
            mask = to_cpu(dxb) != 0
            assert abs(float(mask.float().mean()) - (1.0 - drop)) < 0.03, float(mask.float().mean())
            zd = torch.where(mask, z / (1.0 - drop), torch.zeros(()))
            close(yg, act1(zd), what='activated output under the regenerated dropout mask')
            close(y2g, torch.relu(zd), what='second activated output under the regenerated dropout mask')
        if ref_raw.shape[1] % 8:
            ld = y.stride(3)
            full = torch.as_strided(y, (ref_raw.shape[0], ld, ref_raw.shape[2], ref_raw.shape[3]), y.stride())
            assert float(full[:, ref_raw.shape[1]:].float().abs().max()) == 0.0
    finally:
        lib.gcc_set_option(_lib.OPT_FUSE_BN, -1)


FINALIZE_CASES = [
    # N, H, W, Ci, Co, k, stride, dgrad        route of the launch that writes the statistic rows
    (16, 64, 64, 128, 256, 4, 2, False),       # halo kernel, stride 2 (PatchGAN L2 at a quarter of its size): 64 rows, 4 groups
    (16, 32, 32, 256, 512, 4, 1, False),       # halo kernel, stride 1 (L4-like, padded grid): 64 rows x 2 column tiles
    (16, 64, 64, 32, 64, 4, 2, False),         # igemm 128-pixel tiles (student d1 at a quarter): 128 rows
    (3, 10, 14, 40, 72, 4, 2, False),          # one ragged tile: a single group, the group's last is the launch's last
    (16, 8, 8, 256, 256, 4, 2, False),         # split-K layer: partial tiles + splitk_fold_stats_kernel
    (16, 4, 4, 512, 512, 4, 2, True),          # split-K, ConvTranspose form (4 phases)
    (2, 32, 32, 64, 128, 4, 2, True),          # un-split ConvTranspose form: rows of four phases
    (5, 9, 9, 24, 40, 4, 2, True),             # odd size: phases of different sizes, early-exit workgroups hold tickets too
]


@pytest.mark.parametrize('case', FINALIZE_CASES)
def test_bn_finalize_in_conv_matches_separate_launch(case, monkeypatch):
    _finalize_in_conv_case(case, monkeypatch)


def test_bn_finalize_in_conv_pair_split_unequal_phases(monkeypatch, conv_plan):
    """a data gradient with BatchNorm statistics on the pair-split 256x256 plan whose stride-2 phases hold different
    numbers of tiles (23 x 23: phases of 2, 2, 2 and 1 row tiles) -- both K halves of the empty tile reach the early exit and only
    one of them may take a ticket, else the finalize fires early and the ticket words stay non-zero for the next launch on the
    same tail workspace (three launches in a row here)."""
    from gcc_amd import _lib
    case = (2, 23, 23, 256, 768, 4, 2, True)
    conv_plan('tile256x256_pair')
    assert _tiles(case[:7] + (1,))[1] == 256256, 'the case must run 256x256 tiles'
    d = _lib.conv_t(2, 23, 23, 256, 768, 4, 4, 2, 1, 256, 0, 768, 0)
    assert _lib.load().gcc_conv_workspace(C.byref(d), 1) > 256 * 256 * 4, 'pair split not planned for this geometry'
    _finalize_in_conv_case(case, monkeypatch)


def _finalize_in_conv_case(case, monkeypatch):
    """round 4: with gcc_epilogue_t.bn the BatchNorm behind a conv is finalized by the last-arriving workgroups of the launch
    that writes the statistic rows (stats_tail).  Same canonical summation order as gcc_bn_finalize: coefficients, saved
    statistics and running statistics must be BIT-identical with the separate launch over the same rows (GCC_IN_CONV_FINALIZE=0
    hands the conv no tail workspace), on every route, launch after launch on one workspace (the ticket words reset themselves);
    and right against torch's batch statistics of the conv output."""
    import torch.nn as nn
    ops = _ops()
    N, H, W, Ci, Co, k, s, dgrad = case
    g = torch.Generator().manual_seed(N + H + Ci)
    Ho, Wo = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    wgt = rb(torch.randn(Co, Ci, k, k, generator=g) * 0.05)
    wp, wtp = ops.pack_weights(master_cl(wgt))
    if not dgrad:
        x = to_dev(rb(torch.randn(N, Ci, H, W, generator=g)))
        Cn, shape = Co, (N, Co, Ho, Wo)
    else:
        x = to_dev(rb(torch.randn(N, Co, Ho, Wo, generator=g)))
        Cn, shape = Ci, (N, Ci, H, W)
    count = shape[0] * shape[2] * shape[3]

    def run(in_conv, reps):
        monkeypatch.setattr(ops, 'IN_CONV_FINALIZE', in_conv)
        bn = nn.BatchNorm2d(Cn).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(1 + 0.1 * torch.randn(Cn, generator=torch.Generator().manual_seed(1)))
            bn.bias.copy_(torch.randn(Cn, generator=torch.Generator().manual_seed(2)))
        st = ops.BNState(Cn, DEV)
        out = ops.new_act(*shape, DEV)
        launches = []
        for _ in range(reps):
            ops.lib().gcc_launch_count(1)
            d = ops.bn_desc(bn, st, count, DEV)
            if not dgrad:
                ops.conv_fprop(x, wp, Co, k, s, 1, out=out, want_stats=True, bn=d)
            else:
                ops.conv_dgrad(x, wtp, Ci, H, W, k, s, 1, out=out, want_stats=True, bn=d)
            launches.append(int(ops.lib().gcc_launch_count(1)))
        torch.cuda.synchronize()
        return out, [t.clone() for t in (st.mean, st.rstd, st.scale, st.shift, bn.running_mean, bn.running_var)], launches
    out1, a, la = run(True, 3)
    out0, b, lb = run(False, 3)
    assert la[0] == lb[0] - 1 and la == [la[0]] * 3, (la, lb)          # one launch fewer, every time
    assert torch.equal(out1, out0)
    for name, u, v in zip(('mean', 'rstd', 'scale', 'shift', 'running_mean', 'running_var'), a, b):
        assert torch.equal(u, v), name
    raw = to_cpu(out1)
    close(a[0].cpu(), raw.mean((0, 2, 3)), tol=1e-4, floor=1e-5, what='batch mean')
    close(a[1].cpu(), 1.0 / torch.sqrt(raw.var((0, 2, 3), unbiased=False) + 1e-5), tol=1e-4, floor=1e-5, what='rstd')


@pytest.mark.parametrize('N,H,W,Ci,Co', [(3, 37, 29, 6, 72), (2, 32, 32, 3, 32), (1, 64, 64, 5, 200), (2, 16, 16, 64, 128)])
@pytest.mark.parametrize('mode', ['relu', 'gate'])
def test_conv_second_output(N, H, W, Ci, Co, mode, monkeypatch):
    """gcc_epilogue_t.y2 (round 4): the image-layer convs write a second output from the same launch -- relu(out) (the U-Net's
    first skip) or out * gate[c] (the masked PatchGAN's first DifferentiableOP) -- bit-identical with the separate gcc_bnact_fwd
    launch it replaces; geometries the thin route does not take (last case) fall back to that launch inside ops.conv_fprop."""
    ops = _ops()
    g = torch.Generator().manual_seed(N * H + Co)
    x = to_dev(rb(torch.randn(N, Ci, H, W, generator=g)))
    wgt = rb(torch.randn(Co, Ci, 4, 4, generator=g) * 0.2)
    b = torch.randn(Co, generator=g).to(DEV)
    wp, _ = ops.pack_weights(master_cl(wgt))
    gate = (torch.randint(0, 3, (Co,), generator=g).float() * 0.5).to(DEV)
    Ho, Wo = (H + 2 - 4) // 2 + 1, (W + 2 - 4) // 2 + 1
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, 'CONV_Y2', fused)
        y, y2 = ops.new_act(N, Co, Ho, Wo, DEV), ops.new_act(N, Co, Ho, Wo, DEV)
        ops.lib().gcc_launch_count(1)
        if mode == 'relu':
            ops.conv_fprop(x, wp, Co, 4, 2, 1, out=y, bias=b, act=ops.ACT_LRELU, y2=y2, y2_mode=ops.Y2_RELU)
        else:
            ops.conv_fprop(x, wp, Co, 4, 2, 1, out=y, bias=b, act=ops.ACT_LRELU, y2=y2, y2_mode=ops.Y2_GATE, y2_gate=gate)
        outs.append((y.clone(), y2.clone(), int(ops.lib().gcc_launch_count(1))))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    thin = Ci <= 8
    if thin:
        assert outs[0][2] == 1 and outs[1][2] == 2, (outs[0][2], outs[1][2])
    else:                       # the library's own fallback: the conv's launches, then the gcc_bnact_fwd launch -- either way
        assert outs[0][2] == outs[1][2] >= 2, (outs[0][2], outs[1][2])
    yf = to_cpu(outs[0][0])
    ref2 = torch.relu(yf) if mode == 'relu' else rb(yf * gate.cpu()[None, :, None, None])
    assert torch.equal(to_cpu(outs[0][1]), ref2)
    ref = F.leaky_relu(F.conv2d(to_cpu(x), wgt, b.cpu(), stride=2, padding=1), 0.2)
    close(yf, ref, what='first output')


def test_conv_transpose_as_dgrad_with_stats_and_tanh():
    """ConvTranspose2d(k4,s2,p1) forward == gcc_conv_dgrad of the adjoint conv; weight layout
    [Cin_T, Cout_T, k, k] channels_last is the adjoint conv's [Co][tap][Ci]."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    N, Cin, Cout, h = 2, 64, 24, 8
    x = rb(torch.randn(N, Cin, h, h, generator=g))
    w = rb(torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1)
    b = torch.randn(Cout, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv_transpose2d(xr, wr, None, stride=2, padding=1)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    m = master_cl(w)
    wp, wtp = ops.pack_weights(m)         # adjoint conv: Co_eq = Cin, Ci_eq = Cout
    xd = to_dev(x)
    y, stats = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * h, 4, 2, 1, want_stats=True)
    yg = to_cpu(y)
    close(yg, y_ref.detach(), what='convT fwd')
    st = stats.sum(0).cpu()
    close(st[0], yg.sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='convT stats')
    close(st[1], (yg * yg).sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='convT stats sq')
    y2 = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * h, 4, 2, 1, bias=b.to(DEV), act=ops.ACT_TANH)
    close(to_cpu(y2), torch.tanh(y_ref.detach() + b[None, :, None, None]), what='convT + bias + tanh')
    # its input gradient is the adjoint conv's fprop; its weight gradient the adjoint's wgrad
    dyd = to_dev(dy)
    dx = ops.conv_fprop(dyd, wp, Cin, 4, 2, 1)
    close(to_cpu(dx), xr.grad, what='convT dgrad')
    dw = torch.zeros_like(m)
    ops.conv_wgrad(dyd, xd, dw, 4, 2, 1)
    close(dw.cpu(), wr.grad, tol=5e-3, floor=1e-4, what='convT wgrad')


@pytest.mark.parametrize('Cin,h,w,Cout,N', [(128, 128, 128, 6, 2), (128, 64, 32, 3, 2), (96, 12, 20, 3, 2), (72, 8, 8, 6, 2),
                                             (128, 16, 16, 6, 40), (128, 9, 12, 3, 100), (128, 32, 32, 6, 16)])
def test_data_gradient_to_image_wide(Cin, h, w, Cout, N):
    """65 .. 128 channels into <= 8 (the first PatchGAN layer's data gradient at ndf 128 -- the true 256 x 256 shape at N = 2 --
    and the teacher generator's last ConvTranspose): the LDS-staged thin kernel, the texture-path / implicit-GEMM route
    (GCC_OPT_IGEMM_THIN = 2) on the same inputs, and fp32 torch"""
    ops = _ops()
    from gcc_amd import _lib
    g = torch.Generator().manual_seed(Cin + h)
    x = rb(torch.randn(N, Cin, h, w, generator=g))          # N > 256 / rows: a workgroup walks several row pairs (3-row ring)
    wgt = rb(torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05)
    b = torch.randn(Cout, generator=g) * 0.1
    y_ref = F.conv_transpose2d(x, wgt, None, stride=2, padding=1)
    _, wtp = ops.pack_weights(master_cl(wgt))
    xd = to_dev(x)
    d = ops.conv_desc(N, 2 * h, 2 * w, Cout, Cin, 4, 2, 1, 8, xd.stride(3))
    outs = {}
    for thin in (1, 2):
        prev = ops.lib().gcc_set_option(_lib.OPT_IGEMM_THIN, thin)
        try:
            assert ops.lib().gcc_conv_route(C.byref(d), 1, None) == (1 if (thin == 1 or Cin <= 64) else 0)
            y = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * w, 4, 2, 1)
            close(to_cpu(y), y_ref, what='data gradient -> image (THIN=%d)' % thin)
            y2 = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * w, 4, 2, 1, bias=b.to(DEV), act=ops.ACT_TANH)
            close(to_cpu(y2), torch.tanh(y_ref + b[None, :, None, None]), what='+ bias + tanh (THIN=%d)' % thin)
            base = torch.empty(0, dtype=torch.bfloat16, device=DEV).set_(y2.untyped_storage()).view(N, 2 * h, 2 * w, 8)
            assert float(base[..., Cout:].float().abs().max()) == 0.0, 'padding channels must stay zero'
            outs[thin] = to_cpu(y)
        finally:
            ops.lib().gcc_set_option(_lib.OPT_IGEMM_THIN, prev)
    close(outs[1], outs[2], tol=1e-2, what='staged kernel against the implicit GEMM')


@pytest.mark.parametrize('Cin,h,w', [(128, 16, 16), (64, 10, 24), (24, 6, 6), (40, 20, 9)])
def test_conv_transpose_to_image(Cin, h, w):
    """the generators' last layer: ConvTranspose2d(Cin -> 3, k4 s2 p1) + bias + tanh on the thin-output kernel"""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    N, Cout = 2, 3
    x = rb(torch.randn(N, Cin, h, w, generator=g))
    wgt = rb(torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1)
    b = torch.randn(Cout, generator=g) * 0.1
    y_ref = F.conv_transpose2d(x, wgt, None, stride=2, padding=1)
    _, wtp = ops.pack_weights(master_cl(wgt))
    xd = to_dev(x)
    y = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * w, 4, 2, 1)
    close(to_cpu(y), y_ref, what='convT -> image')
    y2 = ops.conv_dgrad(xd, wtp, Cout, 2 * h, 2 * w, 4, 2, 1, bias=b.to(DEV), act=ops.ACT_TANH)
    close(to_cpu(y2), torch.tanh(y_ref + b[None, :, None, None]), what='convT -> image + bias + tanh')
    base = torch.empty(0, dtype=torch.bfloat16, device=DEV).set_(y2.untyped_storage()).view(N, 2 * h, 2 * w, 8)
    assert float(base[..., Cout:].float().abs().max()) == 0.0, 'padding channels must stay zero'


def test_conv_stats_with_bias_on_partial_tiles():
    """BatchNorm partial statistics of a biased conv whose M tile is only partly filled (SAGAN generator: spectrally
    normalised ConvTranspose2d with bias in front of a BatchNorm): the padding rows must not contribute the bias"""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    x = rb(torch.randn(4, 64, 4, 4, generator=g))
    w = rb(torch.randn(64, 32, 4, 4, generator=g) * 0.1)
    b = torch.randn(32, generator=g)
    wp, wtp = ops.pack_weights(master_cl(w))
    y, stats = ops.conv_dgrad(to_dev(x), wtp, 32, 8, 8, 4, 2, 1, bias=b.to(DEV), want_stats=True)
    yg = to_cpu(y)
    close(yg, F.conv_transpose2d(x, w, b, stride=2, padding=1), what='biased convT')
    st = stats.sum(0).cpu()
    close(st[0], yg.sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='stats sum (bias, partial tile)')
    close(st[1], (yg * yg).sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='stats sumsq (bias, partial tile)')
    w2 = rb(torch.randn(24, 16, 3, 3, generator=g) * 0.1)
    b2 = torch.randn(24, generator=g)
    x2 = rb(torch.randn(1, 16, 9, 9, generator=g))
    wp2, _ = ops.pack_weights(master_cl(w2))
    y2, st2 = ops.conv_fprop(to_dev(x2), wp2, 24, 3, 1, 1, bias=b2.to(DEV), want_stats=True)
    y2g = to_cpu(y2)
    close(st2.sum(0).cpu()[0], y2g.sum((0, 2, 3)), tol=1e-3, floor=1e-3, what='fprop stats sum (bias, partial tile)')


def test_conv_channel_slices():
    """conv reading from / writing into channel slices of wider (concat) buffers"""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    x = rb(torch.randn(2, 32, 8, 8, generator=g))
    w = rb(torch.randn(16, 32, 4, 4, generator=g) * 0.1)
    big_in = ops.new_act(2, 64, 8, 8, DEV)
    big_in[:, 32:64].copy_(x.bfloat16().to(DEV))
    big_out = ops.new_act(2, 48, 4, 4, DEV)
    wp, _ = ops.pack_weights(master_cl(w))
    ops.conv_fprop(ops.cslice(big_in, 32, 32), wp, 16, 4, 2, 1, out=ops.cslice(big_out, 16, 16))
    ref = F.conv2d(x, w, None, stride=2, padding=1)
    close(to_cpu(big_out[:, 16:32]), ref, what='slice conv')
    assert float(big_out[:, :16].float().abs().max()) == 0.0 and float(big_out[:, 32:].float().abs().max()) == 0.0


@pytest.mark.parametrize('fused', [1, 0])
def test_pack_plan_matches_single_tensor_packing(fused, monkeypatch):
    """ops.PackPlan (one launch for a list of convs; with GCC_FUSED_PACK the W and Wt of a 64x64 tile come from one read of the
    master) against gcc_pack_weights per tensor and against the definition: widths that are / are not multiples of 64, 8, 4;
    a padded row count; a concatenated column dimension (keeps the two-kind path either way)"""
    ops = _ops()
    from gcc_amd import engine
    monkeypatch.setattr(ops, 'FUSED_PACK', bool(fused))
    g = torch.Generator().manual_seed(4)
    convs = []
    for rows, cols, k, csplit in ((64, 32, 4, 0), (130, 68, 3, 0), (256, 512, 4, 0), (24, 8, 1, 0), (3, 64, 4, 0), (40, 6, 3, 0),
                                  (32, 20, 4, 12), (200, 128, 1, 0)):
        w = (torch.randn(rows, cols, k, k, generator=g) * 0.1).to(DEV).contiguous(memory_format=torch.channels_last)
        convs.append(engine.ConvOp(w, None, k, 1, 0, False, col_split=csplit))
    plan = ops.PackPlan(convs, DEV)
    kinds = set(int(v) for v in plan.d_items.cpu()[:, 1])
    assert kinds == ({0, 1, 2} if fused else {0, 1})
    plan.run()
    torch.cuda.synchronize()
    for c in convs:
        m = c.weight.detach().float().cpu()                  # [rows, cols, k, k]
        rows, cols, k = c.rows, c.cols, c.k
        ref = m.permute(0, 2, 3, 1).reshape(rows, k * k, cols).to(torch.bfloat16)
        if not c.col_split:
            w1, wt1 = ops.pack_weights(c.weight)
            assert torch.equal(c.w.cpu()[:rows], w1.cpu()) and torch.equal(c.wt.cpu()[:cols], wt1.cpu()), (rows, cols, k)
            assert torch.equal(c.w.cpu()[:rows, :, :cols], ref)
            assert torch.equal(c.wt.cpu()[:cols, :, :rows], ref.permute(2, 1, 0))
            wz, wtz = c.w.cpu().float(), c.wt.cpu().float()                     # padding rows / columns are zeros
            assert not wz[rows:].any() and not wz[:, :, cols:].any() and not wtz[cols:].any() and not wtz[:, :, rows:].any()
        else:
            s0 = c.col_split
            p0 = (s0 + 7) // 8 * 8
            assert torch.equal(c.w.cpu()[:rows, :, :s0], ref[:, :, :s0])
            assert torch.equal(c.w.cpu()[:rows, :, p0:p0 + cols - s0], ref[:, :, s0:])


def test_layout_roundtrip_and_copy():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    a = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    b = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    ab = ops.new_act(2, 6, 16, 16, DEV)
    ops.nchw_to_nhwc(a.to(DEV), ab, 0)
    ops.nchw_to_nhwc(b.to(DEV), ab, 3, cfill=5)
    close(to_cpu(ab), rb(torch.cat([a, b], 1)), tol=0, floor=0, what='pack AB')
    back = ops.nhwc_to_nchw(ab)
    close(back.cpu(), rb(torch.cat([a, b], 1)), tol=0, floor=0, what='unpack')
    f = to_dev(a)
    ops.nhwc_copy(f, 0, ab, 3, 3, cfill=5)
    close(to_cpu(ab), rb(torch.cat([a, a], 1)), tol=0, floor=0, what='copy slice')
    ops.nhwc_add(f, 0, ab, 3, 3)
    close(to_cpu(ab[:, 3:6]), rb(rb(a) + rb(a)), tol=0, floor=0, what='add slice')
    # the discriminator's conditional input: cat(A, B) as one 16-byte pack, also into the second group of a wider buffer
    fa, fb = to_dev(a), to_dev(b)
    pk = ops.new_act(2, 6, 16, 16, DEV)
    base = torch.empty(0, dtype=torch.bfloat16, device=DEV).set_(pk.untyped_storage()).view(2, 16, 16, 8)
    base.fill_(1.0)
    ops.nhwc_pack_pair(fa, fb, pk, 3, 3)
    close(to_cpu(pk), rb(torch.cat([a, b], 1)), tol=0, floor=0, what='pack pair')
    assert float(base[..., 6:].float().abs().max()) == 0.0, 'pack pair zero-fills the group'
    wide = ops.new_act(2, 16, 16, 16, DEV)
    ops.nhwc_pack_pair(fb, fa, wide, 3, 2, doff=8)
    close(to_cpu(wide[:, 8:13]), rb(torch.cat([b, a[:, :2]], 1)), tol=0, floor=0, what='pack pair at offset 8')
    assert float(wide[:, :8].float().abs().max()) == 0.0 and float(wide[:, 13:].float().abs().max()) == 0.0
    whole = ops.new_act(2, 3, 16, 16, DEV)
    basew = torch.empty(0, dtype=torch.bfloat16, device=DEV).set_(whole.untyped_storage()).view(2, 16, 16, 8)
    basew.fill_(1.0)
    ops.nhwc_copy(fa, 0, whole, 0, 3, cfill=8)
    close(to_cpu(whole), rb(a), tol=0, floor=0, what='copy whole group')
    assert float(basew[..., 3:].float().abs().max()) == 0.0
    x = torch.randn(2, 64, 4, 4, generator=g)
    d1, d2 = to_dev(x), to_dev(x)
    ops.nhwc_add(d1, 0, d2, 0, 64)
    close(to_cpu(d2), rb(2 * rb(x)), tol=0, floor=0, what='vector add')


@pytest.mark.parametrize('C,gate,after,drop', [(64, False, False, 0.0), (128, True, False, 0.0), (24, True, True, 0.0),
                                               (256, False, False, 0.5)])
def test_bn_act_gate_forward_backward(C, gate, after, drop):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    N, H, W = 4, 6, 5
    x = rb(torch.randn(N, C, H, W, generator=g) * 2 + 0.5)
    gamma = (1 + 0.1 * torch.randn(C, generator=g))
    beta = torch.randn(C, generator=g)
    alpha = torch.rand(C, generator=g)
    alpha[0] = 0.5
    tau = 0.5
    mask = (torch.sign(alpha - tau) + 1) / 2
    bn = not after       # the gate-after-activation layer (PatchGAN L1) has no norm
    # --- device forward: stats by a 1x1 identity-free path: use channel sums of x and x^2 via torch on CPU
    xd = to_dev(x)
    st = ops.BNState(C, DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    if bn:
        cnt = N * H * W
        stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))])[None].contiguous().to(DEV)
        ops.bn_finalize(stats, cnt, gamma.to(DEV), beta.to(DEV), rm, rv, st)
    md = torch.empty(C, device=DEV)
    ops.gate_mask(alpha.to(DEV), tau, md)
    torch.cuda.synchronize()
    assert torch.equal(md.cpu(), mask)
    y = ops.new_act(N, C, H, W, DEV)
    y2 = ops.new_act(N, C, H, W, DEV)
    ops.bnact_fwd(xd, y, y2, scale=st.scale if bn else None, shift=st.shift if bn else None,
                  gate=md if gate else None, gate_after_act=after, act=ops.ACT_LRELU, act2=ops.ACT_RELU,
                  drop_p=drop, seed=1234)
    # --- CPU reference
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ar = alpha.clone().requires_grad_(True)
    if bn:
        z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    else:
        z = xr
    yg = to_cpu(y)
    if drop > 0:
        # recover the device's keep mask from its own output (z != 0 almost surely)
        keep = (yg != 0).float()
        frac = keep.mean().item()
        assert abs(frac - (1 - drop)) < 0.03, frac
        z = z * keep / (1 - drop)

    class M(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return (torch.sign(a - tau) + 1) / 2

        @staticmethod
        def backward(ctx, gg):
            return gg

    mk = M.apply(ar)[None, :, None, None] if gate else 1.0
    if after:
        y_ref = F.leaky_relu(z, 0.2) * mk
        y2_ref = F.relu(z * mk)
    else:
        y_ref = F.leaky_relu(z * mk, 0.2)
        y2_ref = F.relu(z * mk)
    close(yg, y_ref.detach(), what='bnact y')
    close(to_cpu(y2), y2_ref.detach(), what='bnact y2')
    if bn:
        torch.cuda.synchronize()
        close(st.mean.cpu(), x.mean((0, 2, 3)), tol=1e-4, floor=1e-5, what='mean')
        close(rv.cpu(), 0.9 + 0.1 * x.var((0, 2, 3), unbiased=True), tol=1e-3, what='running var')
    # --- backward
    g1 = rb(torch.randn(N, C, H, W, generator=g))
    g2 = rb(torch.randn(N, C, H, W, generator=g)) if not after else None
    loss = (y_ref * g1).sum() + ((y2_ref * g2).sum() if g2 is not None else 0.0)
    loss.backward()
    dgamma, dbeta, dalpha = (torch.zeros(C, device=DEV) for _ in range(3))
    dx = ops.new_act(N, C, H, W, DEV)
    ops.bnact_bwd(xd, y, to_dev(g1), dx, g2=to_dev(g2) if g2 is not None else None, bn=st if bn else None,
                  gamma=gamma.to(DEV) if bn else None, beta=beta.to(DEV) if bn else None, gate=md if gate else None,
                  gate_after_act=after, act=ops.ACT_LRELU, act2=ops.ACT_RELU, drop_p=drop, seed=1234,
                  dgamma=dgamma if bn else None, dbeta=dbeta if bn else None, dalpha=dalpha if gate else None)
    close(to_cpu(dx), xr.grad, tol=2e-2, what='bnact dx')
    if bn:
        close(dgamma.cpu(), gr.grad, tol=1e-2, floor=1e-3, what='dgamma')
        close(dbeta.cpu(), br.grad, tol=1e-2, floor=1e-3, what='dbeta')
    if gate:
        close(dalpha.cpu(), ar.grad, tol=1e-2, floor=1e-3, what='dalpha')


@pytest.mark.parametrize('C,N,H,W,drop,noy,two', [(20, 3, 5, 7, 0.0, False, True), (256, 16, 16, 16, 0.0, True, True),
                                                  (64, 16, 8, 8, 0.5, True, False), (512, 16, 2, 2, 0.5, False, True),
                                                  (32, 5, 29, 29, 0.0, True, True)])
def test_bn_backward_small_tensor_one_launch(C, N, H, W, drop, noy, two, monkeypatch):
    """gcc_bnact_bwd of the U-Net's layers as ONE kernel -- gcc_bn_bwd_one_launch_ex (round 4: the grid kernel on the whole chip,
    any size) or bnact_bwd_small_kernel (<= 4096 pixels, GCC_OPT_BN_BWD_SMALL) -- against fp32 torch autograd and against the
    three-launch pipeline on the same inputs (dropout mask regenerated from the same counter; y given or recomputed from x
    through the forward's affine; one or two incoming gradients).  The last case (4205 pixels) is above the small kernel's limit
    and must take the three-launch path under both of its settings."""
    ops = _ops()
    from gcc_amd import _lib
    g = torch.Generator().manual_seed(C + H)
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), torch.randn(C, generator=g)
    xd = to_dev(x)
    st = ops.BNState(C, DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))])[None].contiguous().to(DEV)
    ops.bn_finalize(stats, N * H * W, gamma.to(DEV), beta.to(DEV), rm, rv, st)
    y, y2 = ops.new_act(N, C, H, W, DEV), ops.new_act(N, C, H, W, DEV)
    ops.bnact_fwd(xd, y, y2, scale=st.scale, shift=st.shift, act=ops.ACT_LRELU, act2=ops.ACT_RELU, drop_p=drop, seed=77)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    if drop > 0:
        keep = (to_cpu(y) != 0).float()
        z = z * keep / (1 - drop)
    g1 = rb(torch.randn(N, C, H, W, generator=g))
    g2 = rb(torch.randn(N, C, H, W, generator=g)) if two else None
    loss = (F.leaky_relu(z, 0.2) * g1).sum() + ((F.relu(z) * g2).sum() if two else 0.0)
    loss.backward()
    res = {}
    for small in (2, 1, 0):         # 2: the extended grid kernel; 1: the small-tensor kernel; 0: three launches
        monkeypatch.setattr(ops, 'BN_BWD_GRID_EX', 2 if small == 2 else 0)
        prev = ops.lib().gcc_set_option(_lib.OPT_BN_BWD_SMALL, 1 if small else 0)
        try:
            dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            dx = ops.new_act(N, C, H, W, DEV)
            ops.lib().gcc_launch_count(1)
            ops.bnact_bwd(xd, None if noy else y, to_dev(g1), dx, g2=to_dev(g2) if two else None, bn=st, gamma=gamma.to(DEV),
                          beta=beta.to(DEV), act=ops.ACT_LRELU, act2=ops.ACT_RELU, drop_p=drop, seed=77, dgamma=dgamma, dbeta=dbeta)
            launches = int(ops.lib().gcc_launch_count(1))
            res[small] = (to_cpu(dx), dgamma.cpu(), dbeta.cpu())
        finally:
            ops.lib().gcc_set_option(_lib.OPT_BN_BWD_SMALL, prev)
        if small == 2:
            assert launches == 1, launches
        close(res[small][0], xr.grad, tol=2e-2, what='dx (route %d)' % small)
        close(res[small][1], gr.grad, tol=1e-2, floor=1e-3, what='dgamma (route %d)' % small)
        close(res[small][2], br.grad, tol=1e-2, floor=1e-3, what='dbeta (route %d)' % small)
    close(res[2][0], res[0][0], tol=1e-2, what='dx, grid kernel against three launches')
    close(res[2][1], res[0][1], tol=1e-4, floor=1e-4, what='dgamma, grid kernel against three launches')
    # the two paths against each other: same sums up to their order, dz unrounded in the one-launch kernel
    close(res[1][0], res[0][0], tol=1e-2, what='dx, one launch against three')
    close(res[1][1], res[0][1], tol=1e-4, floor=1e-4, what='dgamma, one launch against three')
    if N * H * W > 4096:
        assert torch.equal(res[1][0], res[0][0])


@pytest.mark.parametrize('N,C,H,W,gate,two', [(16, 64, 64, 64, False, True), (16, 256, 32, 32, True, False), (8, 40, 33, 17, False, False),
                                              (16, 128, 128, 128, True, False)])
def test_bn_backward_finalize_inside_the_reduce_launch(N, C, H, W, gate, two, monkeypatch):
    """round 4: gcc_bnact_bwd with flags bit 0 (zero-filled, stream-private workspace) lets the reduce pass's last-arriving
    workgroups do the finalize step (bwd_tail): two launches instead of three, the same gradients (fp32 sums in another order:
    1e-5 relative), bit-identical from call to call on one workspace (the ticket words reset themselves)"""
    ops = _ops()
    g = torch.Generator().manual_seed(N + C)
    x = to_dev(rb(torch.randn(N, C, H, W, generator=g) * 1.3 + 0.2))
    g1 = to_dev(rb(torch.randn(N, C, H, W, generator=g)))
    g2 = to_dev(rb(torch.randn(N, C, H, W, generator=g))) if two else None
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    mask = (torch.randint(0, 3, (C,), generator=g).float() * 0.5).to(DEV) if gate else None
    st = ops.BNState(C, DEV)
    xf = x.float()
    st.mean.copy_(xf.mean((0, 2, 3))); st.rstd.copy_(1.0 / torch.sqrt(xf.var((0, 2, 3), unbiased=False) + 1e-5))
    st.scale.copy_(gamma * st.rstd); st.shift.copy_(beta - st.mean * gamma * st.rstd)
    outs = []
    for tail in (True, True, False):
        monkeypatch.setattr(ops, 'BN_BWD_TAIL', tail)
        dx = ops.new_act(N, C, H, W, DEV)
        dg, db, da = (torch.zeros(C, device=DEV) for _ in range(3))
        ops.lib().gcc_launch_count(1)
        ops.bnact_bwd(x, None, g1, dx, g2=g2, bn=st, gamma=gamma, beta=beta, gate=mask, act=ops.ACT_LRELU, act2=ops.ACT_RELU,
                      dgamma=dg, dbeta=db, dalpha=da if gate else None)
        outs.append((dx.clone(), dg.clone(), db.clone(), da.clone(), int(ops.lib().gcc_launch_count(1))))
    torch.cuda.synchronize()
    assert outs[0][4] == 2 and outs[2][4] == 3, (outs[0][4], outs[2][4])
    for u, v in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(u, v)
    for name, u, v in zip(('dgamma', 'dbeta', 'dalpha'), outs[0][1:4], outs[2][1:4]):
        assert torch.allclose(u, v, rtol=1e-5, atol=1e-4 * max(1.0, float(v.abs().max()))), name
    d = (outs[0][0].float() - outs[2][0].float()).abs().max()
    assert float(d) <= 1e-2 * max(1.0, float(outs[2][0].float().abs().max())), float(d)


@pytest.mark.parametrize('N,C,H,W', [(3, 40, 7, 5), (1, 24, 128, 128), (1, 64, 129, 128), (2, 8, 64, 100), (16, 256, 1, 1)])
def test_channel_sum(N, C, H, W):
    """bias gradients: the one-launch kernel (<= 16384 pixels) and the two-launch pipeline, plain and accumulating"""
    ops = _ops()
    from gcc_amd import _lib
    x = rb(torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(C + H)))
    want = x.double().sum((0, 2, 3)).float()
    xd = to_dev(x)
    for small in (1, 0):
        prev = ops.lib().gcc_set_option(_lib.OPT_BN_BWD_SMALL, small)
        try:
            out = torch.zeros(C, device=DEV)
            ops.channel_sum(xd, out)
            close(out.cpu(), want, tol=1e-4, floor=2e-3, what='channel sum (one launch: %d)' % small)
            ops.channel_sum(xd, out, accumulate=True)
            close(out.cpu(), 2 * want, tol=1e-4, floor=4e-3, what='accumulated channel sum (one launch: %d)' % small)
        finally:
            ops.lib().gcc_set_option(_lib.OPT_BN_BWD_SMALL, prev)


@pytest.mark.parametrize('mode', ['hinge', 'lsgan', 'vanilla', 'wgangp'])
def test_gan_loss(mode):
    ops = _ops()
    from oracle import gcc_oracle as O
    g = torch.Generator().manual_seed(3)
    pred = rb(torch.randn(2, 1, 30, 30, generator=g) * 1.5)
    pd = to_dev(pred)
    for real in (True, False):
        for ford in (True, False):
            if mode == 'hinge' and not ford and not real:
                continue
            pr = pred.clone().requires_grad_(True)
            l = O.gan_loss(mode, pr, real, ford)
            (l * 0.5).backward()
            loss = torch.zeros(1, device=DEV)
            dp = ops.new_act(2, 1, 30, 30, DEV)
            ops.gan_loss(mode, pd, real, ford, loss, dpred=dp, grad_weight=0.5)
            assert abs(loss.item() - l.item()) < 1e-4 * max(1, abs(l.item())), (mode, real, ford)
            close(to_cpu(dp), pr.grad, tol=1e-2, floor=1e-9, what='dpred %s' % mode)
            wd = torch.full((1,), -2.0, device=DEV)
            ops.gan_loss(mode, pd, real, ford, loss, dpred=dp, grad_weight=0.5, weight_dev=wd, dpred_accumulate=True)
            close(to_cpu(dp), -pr.grad, tol=2e-2, floor=1e-9, what='dpred accumulate %s' % mode)


def test_l1_loss():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    a = rb(torch.rand(2, 3, 32, 32, generator=g) * 2 - 1)
    b = rb(torch.rand(2, 3, 32, 32, generator=g) * 2 - 1)
    ar = a.clone().requires_grad_(True)
    l = F.l1_loss(ar, b) * 100.0
    l.backward()
    loss = torch.zeros(1, device=DEV)
    da = ops.new_act(2, 3, 32, 32, DEV)
    ops.l1_loss(to_dev(a), to_dev(b), loss, weight=100.0, da=da)
    assert abs(loss.item() - l.item()) < 1e-4 * l.item()
    close(to_cpu(da), ar.grad, tol=1e-2, floor=1e-9, what='l1 grad')


@pytest.mark.parametrize('squared', [False, True])
@pytest.mark.parametrize('N,C,H,W', [(2, 128, 8, 8), (2, 256, 5, 5), (1, 64, 16, 16)])
def test_distill_loss(N, C, H, W, squared):
    """gram + content terms: RMSE form (Pix2Pix) and plain MSE form (CycleGAN)"""
    ops = _ops()
    from oracle import gcc_oracle as O
    g = torch.Generator().manual_seed(C)
    f = rb(torch.randn(N, C, H, W, generator=g))
    t = rb(torch.randn(N, C, H, W, generator=g) * 0.8 + 0.1)
    fr = f.clone().requires_grad_(True)
    if squared:
        lg, lc = F.mse_loss(O.gram(fr), O.gram(t)), F.mse_loss(fr, t)
    else:
        lg, lc = O.rmse(O.gram(fr), O.gram(t)), O.rmse(fr, t)
    wg, wc = 1e4, 50.0
    (wg * lg + wc * lc).backward()
    ws = torch.empty(ops.distill_workspace_bytes(N, C, H * W), dtype=torch.uint8, device=DEV)
    out = torch.zeros(2, device=DEV)
    fd, td = to_dev(f), to_dev(t)
    ops.distill_fwd(fd, td, out, ws, squared=squared)
    o = out.cpu()
    assert abs(o[0].item() - lg.item()) < 1e-2 * lg.item(), (o[0].item(), lg.item())
    assert abs(o[1].item() - lc.item()) < 2e-3 * lc.item(), (o[1].item(), lc.item())
    df = ops.new_act(N, C, H, W, DEV)
    ops.distill_bwd(fd, td, wg, wc, df, ws, squared=squared)
    close(to_cpu(df), fr.grad, tol=2e-2, what='distill grad')


def test_adam_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    shapes = [(70000,), (33, 7, 4, 4), (5,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    dp = [p.to(DEV) for p in ps]
    dg = [torch.zeros_like(p) for p in dp]
    plan = ops.AdamPlan(dp, dg, DEV)
    for it in range(3):
        gs = [torch.randn(s, generator=g) for s in shapes]
        for r, gg, d in zip(ref, gs, dg):
            r.grad = gg.clone()
            d.copy_(gg.to(DEV))
        opt.step()
        plan.step(2e-4, (0.5, 0.999))
    for r, d in zip(ref, dp):
        close(d.cpu(), r.detach(), tol=1e-6, floor=1e-7, what='adam')
    # L1 sub-gradient fused
    p = torch.randn(1000, generator=g)
    r = p.clone().requires_grad_(True)
    o2 = torch.optim.Adam([r], lr=1e-3)
    gg = torch.randn(1000, generator=g)
    r.grad = gg + 0.01 * torch.sign(p)
    o2.step()
    d, dgr = p.to(DEV), gg.to(DEV)
    pl = ops.AdamPlan([d], [dgr], DEV, l1=[0.01])
    pl.step(1e-3)
    close(d.cpu(), r.detach(), tol=1e-6, floor=1e-7, what='adam+l1')


def test_bad_arguments_return_errors():
    import ctypes as C
    from gcc_amd import _lib
    lib = _lib.load()
    d = _lib.conv_t(1, 8, 8, 8, 8, 4, 4, 2, 1, 7, 0, 8, 0)       # ldx not a multiple of 8
    assert lib.gcc_conv_fprop(C.byref(d), 1, 1, 1, None, None) == -1
    assert lib.gcc_conv_fprop(None, None, None, None, None, None) == -1
    assert b'workspace' in lib.gcc_strerror(-3)


def test_reflect_pad_and_adjoint():
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    x = rb(torch.randn(2, 16, 9, 7, generator=g))
    for pad in (1, 3):
        xr = x.clone().requires_grad_(True)
        ref = F.pad(xr, (pad,) * 4, mode='reflect')
        gy = rb(torch.randn(ref.shape, generator=g))
        ref.backward(gy)
        out = ops.new_act(2, 16, 9 + 2 * pad, 7 + 2 * pad, DEV)
        ops.reflect_pad(to_dev(x), out, pad)
        close(to_cpu(out), ref.detach(), tol=0, floor=0, what='reflect pad')
        dx = ops.new_act(2, 16, 9, 7, DEV)
        ops.reflect_pad(to_dev(gy), dx, pad, backward=True)
        close(to_cpu(dx), xr.grad, tol=1e-2, what='reflect pad adjoint')


@pytest.mark.parametrize('C,H,W', [(32, 8, 8), (20, 6, 9), (256, 16, 16)])
def test_depthwise_conv_reflect(C, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    N = 2
    x = rb(torch.randn(N, C, H, W, generator=g))
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(C, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode='reflect'), wr, br, groups=C)
    gy = rb(torch.randn(ref.shape, generator=g))
    ref.backward(gy)
    xd, gd = to_dev(x), to_dev(gy)
    wd, bd = w.to(DEV), b.to(DEV)
    y = ops.new_act(N, C, H, W, DEV)
    ops.dwconv_fwd(xd, wd, bd, y)
    close(to_cpu(y), ref.detach(), what='dwconv fwd')
    dx = ops.new_act(N, C, H, W, DEV)
    ops.dwconv_bwd_data(gd, wd, dx)
    close(to_cpu(dx), xr.grad, what='dwconv dgrad')
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    ops.dwconv_wgrad(xd, gd, dw, db)
    close(dw.cpu(), wr.grad, tol=5e-3, floor=1e-3, what='dwconv wgrad')
    close(db.cpu(), br.grad, tol=5e-3, floor=1e-3, what='dwconv bias grad')


@pytest.mark.parametrize('C,relu,res', [(64, True, False), (24, False, True), (256, True, True)])
def test_instance_norm_forward_backward(C, relu, res):
    ops = _ops()
    g = torch.Generator().manual_seed(C + 1)
    N, H, W = 3, 8, 6
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    r = rb(torch.randn(N, C, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    z = F.instance_norm(xr, eps=1e-5)
    yref = F.relu(z) if relu else z
    if res:
        yref = yref + r
    gy = rb(torch.randn(N, C, H, W, generator=g))
    yref.backward(gy)
    xd = to_dev(x)
    st = ops.INState(N, C, DEV)
    ops.in_finalize(ops.channel_stats(xd), H * W, st)
    y = ops.new_act(N, C, H, W, DEV)
    ops.bnact_fwd(xd, y, scale=st.scale, shift=st.shift, act=ops.ACT_RELU if relu else ops.ACT_NONE, groups=N,
                  residual=to_dev(r) if res else None)
    close(to_cpu(y), yref.detach(), what='instance norm fwd')
    close(st.mean.cpu(), x.mean((2, 3)), tol=1e-4, floor=1e-5, what='IN mean')
    dx = ops.new_act(N, C, H, W, DEV)
    # backward of act(IN(x)) : the saved output for act' is y without the residual -> recompute from x (y=None)
    ops.bnact_bwd(xd, None if res else y, to_dev(gy), dx, bn=st, act=ops.ACT_RELU if relu else ops.ACT_NONE, groups=N)
    close(to_cpu(dx), xr.grad, tol=2e-2, what='instance norm bwd')


@pytest.mark.parametrize('grid', [1, 0])
@pytest.mark.parametrize('C,H,W,relu,res', [(12, 8, 6, True, False), (8, 5, 7, False, True), (40, 64, 64, True, False),
                                            (16, 64, 72, False, True), (24, 128, 128, True, False), (256, 64, 64, True, True),
                                            (96, 64, 64, False, False), (512, 32, 31, True, False), (1032, 16, 16, True, False),
                                            (64, 256, 256, True, False), (3, 33, 17, False, False)])
def test_instance_norm_one_launch(C, H, W, relu, res, grid):
    """gcc_inorm_fwd / gcc_inorm_bwd against torch's instance_norm, and against the three-launch pipeline they replace:
    grid=1 the plane of an image split over workgroups that meet at an in-launch barrier (C = 1032 is beyond its plan and
    stays with the slab kernel), grid=0 one workgroup per image x 16-channel slab.  Every call is made twice: the second
    launch finds the first one's tagged partials in the workspace and must not take them for its own."""
    _run_instance_norm_one_launch(C, H, W, relu, res, grid)


def _run_instance_norm_one_launch(C, H, W, relu, res, grid):
    ops = _ops()
    from gcc_amd import _lib
    lib = _lib.load()
    lib.gcc_set_option(_lib.OPT_INORM_GRID, grid)
    try:
        _instance_norm_one_launch_body(ops, C, H, W, relu, res)
    finally:
        lib.gcc_set_option(_lib.OPT_INORM_GRID, -1)


def _instance_norm_one_launch_body(ops, C, H, W, relu, res):
    ops = _ops()
    g = torch.Generator().manual_seed(C + H)
    N = 2
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3)
    r = rb(torch.randn(N, C, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    z = F.instance_norm(xr, eps=1e-5)
    yref = F.relu(z) if relu else z
    if res:
        yref = yref + r
    gy = rb(torch.randn(N, C, H, W, generator=g))
    yref.backward(gy)
    act = ops.ACT_RELU if relu else ops.ACT_NONE
    xd = to_dev(x)
    st = ops.INState(N, C, DEV)
    y = ops.new_act(N, C, H, W, DEV)
    for _ in range(2):
        ops.inorm_fwd(xd, y, st, act=act, residual=to_dev(r) if res else None)
    close(to_cpu(y), yref.detach(), what='one-launch instance norm fwd')
    close(st.mean.cpu(), x.mean((2, 3)), tol=1e-4, floor=1e-5, what='IN mean')
    close(st.rstd.cpu(), 1.0 / torch.sqrt(x.var((2, 3), unbiased=False) + 1e-5), tol=1e-4, what='IN rstd')
    st3 = ops.INState(N, C, DEV)
    ops.in_finalize(ops.channel_stats(xd), H * W, st3)
    assert torch.allclose(st.scale, st3.scale, rtol=4e-6, atol=1e-7) and torch.allclose(st.shift, st3.shift, rtol=4e-5, atol=2e-6)
    gd = to_dev(gy)
    dx = ops.new_act(N, C, H, W, DEV)
    ops.inorm_bwd(xd, None if res else y, gd, dx, st, act=act)
    close(to_cpu(dx), xr.grad, tol=2e-2, what='one-launch instance norm bwd')
    ops.inorm_bwd(xd, None if res else y, gd, gd, st, act=act)          # in place over the incoming gradient
    assert torch.equal(to_cpu(gd), to_cpu(dx))
    if C % 8:                                                           # pad channels stay exact zeros
        assert float(y.permute(0, 2, 3, 1).reshape(-1)[:0].sum()) == 0.0
        base = y.as_strided((N, H, W, ops.ceil8(C)), (H * W * ops.ceil8(C), W * ops.ceil8(C), ops.ceil8(C), 1))
        assert float(base[..., C:].abs().max()) == 0.0


@pytest.mark.parametrize('N,C,H,W,act,with_y', [(16, 64, 24, 24, 'none', False), (16, 24, 24, 24, 'none', False), (64, 96, 16, 16, 'relu', True),
                                                 (2, 256, 96, 96, 'lrelu', True), (16, 520, 24, 24, 'none', False)])
def test_batchnorm_backward_one_launch_vs_three_launches(N, C, H, W, act, with_y, monkeypatch):
    """gcc_bn_bwd_one_launch (the grid InstanceNorm backward with the batch as one plane and gamma in the coefficients) against
    the reduce + finalize + apply route and against fp32 torch autograd: dx, d(gamma) +=, d(beta) +="""
    ops = _ops()
    g = torch.Generator().manual_seed(C + N)
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.2 + 0.1)
    gy = rb(torch.randn(N, C, H, W, generator=g))
    gamma = (torch.rand(C, generator=g) + 0.5)
    beta = torch.randn(C, generator=g) * 0.1
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    yref = {'none': z, 'relu': F.relu(z), 'lrelu': F.leaky_relu(z, 0.2)}[act]
    yref.backward(gy)
    code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'lrelu': ops.ACT_LRELU}[act]
    xd, gd = to_dev(x), to_dev(gy)
    yd = to_dev(rb(yref.detach())) if with_y else None
    st = ops.BNState(C, DEV)
    st.mean.copy_(x.mean((0, 2, 3))); st.rstd.copy_(1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5))
    gam, bet = gamma.to(DEV), beta.to(DEV)
    outs = []
    for grid in (True, False):
        monkeypatch.setattr(ops, 'BN_BWD_GRID', grid)
        dx = ops.new_act(N, C, H, W, DEV)
        dg, db = torch.full((C,), 0.25, device=DEV), torch.full((C,), -0.5, device=DEV)          # accumulated into
        before = ops.lib().gcc_launch_count(1)
        ops.bnact_bwd(xd, yd, gd, dx, bn=st, gamma=gam, beta=bet, act=code, dgamma=dg, dbeta=db)
        launches = ops.lib().gcc_launch_count(1)
        outs.append((to_cpu(dx), dg.cpu() - 0.25, db.cpu() + 0.5, launches))
    assert outs[0][3] == 1 and outs[1][3] == 3, (outs[0][3], outs[1][3])
    close(outs[0][0], xr.grad, tol=2e-2, what='BatchNorm dx (one launch)')
    close(outs[0][1], gr.grad, tol=2e-2, what='d gamma')
    close(outs[0][2], br.grad, tol=2e-2, what='d beta')
    d = (outs[0][0] - outs[1][0]).abs().max()
    assert float(d) <= 2e-2 * max(1.0, float(outs[1][0].abs().max())), float(d)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-3) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-3)


def test_batchnorm_backward_one_launch_needs_the_grid_form(monkeypatch):
    """with GCC_OPT_INORM_GRID = 0 gcc_bn_bwd_one_launch must refuse (GCC_ERR_UNSUPPORTED) instead of running the
    InstanceNorm slab kernel, which knows no gamma / d gamma / d beta; ops.bnact_bwd then takes the three-launch route and the
    gradients are the same as with the grid form"""
    ops = _ops()
    from gcc_amd import _lib
    lib = _lib.load()
    N, C, H, W = 16, 64, 24, 24
    g = torch.Generator().manual_seed(5)
    x = rb(torch.randn(N, C, H, W, generator=g) * 1.2 + 0.1)
    gy = rb(torch.randn(N, C, H, W, generator=g))
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    xd, gd = to_dev(x), to_dev(gy)
    st = ops.BNState(C, DEV)
    st.mean.copy_(x.mean((0, 2, 3))); st.rstd.copy_(1.0 / torch.sqrt(x.var((0, 2, 3), unbiased=False) + 1e-5))
    monkeypatch.setattr(ops, 'BN_BWD_GRID', True)
    outs = []
    for grid in (1, 0):
        lib.gcc_set_option(_lib.OPT_INORM_GRID, grid)
        try:
            dx = ops.new_act(N, C, H, W, DEV)
            dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            lib.gcc_launch_count(1)
            ops.bnact_bwd(xd, None, gd, dx, bn=st, gamma=gamma, beta=beta, act=ops.ACT_NONE, dgamma=dg, dbeta=db)
            outs.append((to_cpu(dx), dg.cpu(), db.cpu(), lib.gcc_launch_count(1)))
            if grid == 0:           # the entry point itself says so
                ws = ops.inorm_workspace(DEV)
                xp, _, _, _, _, ldx = ops.geom(xd)
                gp, _, _, _, _, ldg = ops.geom(gd)
                dxp, _, _, _, _, lddx = ops.geom(dx)
                rc = lib.gcc_bn_bwd_one_launch(xp, ldx, None, 0, gp, ldg, dxp, lddx, C, N * H * W, ops.ACT_NONE, 0.2, st.mean.data_ptr(),
                                               st.rstd.data_ptr(), gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                               ws.numel(), ops.stream())
                assert rc == -2, rc
        finally:
            lib.gcc_set_option(_lib.OPT_INORM_GRID, -1)
    assert outs[0][3] == 1 and outs[1][3] == 3, (outs[0][3], outs[1][3])
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-2 * max(1.0, float(outs[1][0].abs().max()))
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-3) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-3)
    assert float(outs[1][1].abs().max()) > 0.1          # d gamma was really computed


_TIMEOUT_PROBE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from gcc_amd import _lib, ops
lib = _lib.load()
import ctypes as C
lib.gcc_diag_set.restype, lib.gcc_diag_set.argtypes = C.c_int, [C.c_int]
DEV = torch.device('cuda:0')
assert lib.gcc_device_error(1) == 0
x = torch.randn(1, 64, 64, 64, generator=torch.Generator().manual_seed(2)).bfloat16().to(DEV).contiguous(memory_format=torch.channels_last)
y = ops.new_act(1, 64, 64, 64, DEV)
st = ops.INState(1, 64, DEV)
ops.inorm_fwd(x, y, st)
torch.cuda.synchronize()
good = y.clone()
assert lib.gcc_device_error(0) == 0
lib.gcc_diag_set(64)
try:
    ops.inorm_fwd(x, y, st)                     # enqueued fine: the error only exists once the kernel has run
    torch.cuda.synchronize()
finally:
    lib.gcc_diag_set(0)
assert lib.gcc_device_error(0) == 0x1401, hex(lib.gcc_device_error(0))
try:
    ops.inorm_fwd(x, y, st)
    raise SystemExit('no GccError while the error word is set')
except _lib.GccError:
    pass
assert lib.gcc_device_error(1) == 0x1401 and lib.gcc_device_error(0) == 0
ops.inorm_fwd(x, y, st)                         # and the path works again, on the same workspace
torch.cuda.synchronize()
assert torch.equal(y, good)
print('TIMEOUT_PROBE_OK')
'''


def test_instance_norm_exchange_timeout_is_reported():
    """a spin of the grid InstanceNorm's in-launch exchange that expires must not pass silently.  The switch
    that provokes it (workgroup 1 of every domain publishes nothing, polls cut to 256) exists only in the DIAGNOSTIC build of
    the library (libgcc_hip_diag.so, csrc/build.sh -DGCC_DIAG_BUILD: gcc_diag_set(64)) -- the shipped one holds no such branch
     -- so the probe runs in a process of its own that loads that build through GCC_HIP_LIB: the launch
    finishes (its results are wrong), the device error word is set, and the next gcc_inorm_* call returns GCC_ERR_LAUNCH until
    it is cleared."""
    import subprocess
    import sys
    from gcc_amd import _lib
    diag = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libgcc_hip_diag.so')
    assert os.path.exists(diag), 'gcc_amd/csrc/build.sh builds the diagnostic variant beside the library'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _TIMEOUT_PROBE % root], env=dict(os.environ, GCC_HIP_LIB=diag), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and 'TIMEOUT_PROBE_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert _lib.load().gcc_device_error(0) == 0          # this process (the shipped library) saw nothing of it


def test_grid_kernels_on_four_streams_make_progress():
    """the grid InstanceNorm waits for its own workgroups inside the launch; four such launches in flight at
    once (CycleGAN's four chains, one per hardware queue) must all be resident together -- norm_act.hip grid_family_wgs(): four
    workgroups per CU by construction, one quarter of those slots per launch -- so none of them ever waits for a slot that
    another waiting launch holds.  60 rounds of four concurrent forward + backward launches on four streams: every result equals
    the single-stream one bit for bit and the device error word stays clear."""
    ops = _ops()
    lib = ops.lib()
    assert lib.gcc_device_error(1) == 0
    streams = [torch.cuda.Stream() for _ in range(4)]
    xs = [to_dev(rb(torch.randn(1, 64 + 32 * i, 64, 64, generator=torch.Generator().manual_seed(20 + i)))) for i in range(4)]
    gs = [to_dev(rb(torch.randn(1, 64 + 32 * i, 64, 64, generator=torch.Generator().manual_seed(30 + i)))) for i in range(4)]
    ref = []
    for x, g in zip(xs, gs):
        y, dx, st = ops.new_act(*x.shape, DEV), ops.new_act(*x.shape, DEV), ops.INState(1, x.shape[1], DEV)
        ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
        ops.inorm_bwd(x, y, g, dx, st, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        ref.append((y.clone(), dx.clone()))
    outs = [(ops.new_act(*x.shape, DEV), ops.new_act(*x.shape, DEV), ops.INState(1, x.shape[1], DEV)) for x in xs]
    for rnd in range(60):
        for s_, x, g, (y, dx, st) in zip(streams, xs, gs, outs):
            with ops.on_stream(s_):
                ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
                ops.inorm_bwd(x, y, g, dx, st, act=ops.ACT_RELU)
        if rnd % 20 == 19:
            torch.cuda.synchronize()
            assert lib.gcc_device_error(0) == 0, hex(lib.gcc_device_error(0))
            for (y, dx, _), (ry, rdx) in zip(outs, ref):
                assert torch.equal(y, ry) and torch.equal(dx, rdx)


def test_batchnorm_backward_route_by_size(monkeypatch):
    """the one-launch BatchNorm backward is for tensors that are a few launches' worth of latency (<= GCC_BN_BWD_GRID_MAX_BYTES, 12 MB);
    larger ones take reduce + finalize + apply, which runs at the memory system's rate (profiles/r5_bn_bwd_paths.txt)"""
    ops = _ops()
    monkeypatch.setattr(ops, 'BN_BWD_GRID', True)
    for (N, C, H, W), want in (((2, 256, 96, 96), 1), ((4, 256, 96, 96), 3), ((16, 64, 96, 96), 3)):
        x = ops.new_act(N, C, H, W, DEV); x.normal_()
        gy = ops.new_act(N, C, H, W, DEV); gy.normal_()
        y = ops.new_act(N, C, H, W, DEV); y.normal_()
        dx = ops.new_act(N, C, H, W, DEV)
        st = ops.BNState(C, DEV); st.rstd.fill_(1.0)
        gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        ops.lib().gcc_launch_count(1)
        ops.bnact_bwd(x, y, gy, dx, bn=st, gamma=gam, beta=bet, act=ops.ACT_LRELU, dgamma=dg, dbeta=db)
        assert int(ops.lib().gcc_launch_count(1)) == want, (N, C, H, W)


@pytest.mark.parametrize('case', [
    # N, H, W, Ci, Co, k          wide kernels with <= 3 output channels: conv_thinout.hip
    (2, 40, 80, 64, 3, 9),        # SRGAN teacher's last layer shape (64 -> 3, 9 x 9), a strip and a quarter wide, two row bands
    (1, 33, 70, 24, 3, 9),        # student (24 channels: padded to 32 in LDS only), ragged width and height
    (2, 24, 64, 32, 3, 5),        # 5 x 5
    (1, 20, 48, 16, 4, 7),        # k * Co = 28 columns
    (3, 9, 16, 8, 1, 3),          # smallest
    (1, 96, 200, 64, 3, 9),       # many rows: several bands, the ring wraps many times
])
def test_thin_output_wide_kernel_wgrad(case):
    """gcc_conv_wgrad's thin-output route (round 5; the horizontal taps in the MFMA's free dimension, the rows of x through an LDS
    ring, dY expanded along x in LDS) against torch's conv2d_weight on the same bf16-rounded inputs, fresh and accumulating; the
    launch count shows the route was taken (kernel + fold)"""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    dy = rb(torch.randn(N, Co, H, W, generator=g))
    dw_ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), dy, stride=1, padding=pad)
    xd, dyd = to_dev(x), to_dev(dy)
    dw = torch.full((Co, Ci, k, k), 5.0, device=DEV).contiguous(memory_format=torch.channels_last)        # stale contents must go
    ops.lib().gcc_launch_count(1)
    ops.conv_wgrad(xd, dyd, dw, k, 1, pad, accumulate=False)
    assert int(ops.lib().gcc_launch_count(1)) == 2, 'the thin-output route is a kernel + its fold'
    close(dw.cpu(), dw_ref, tol=5e-3, floor=1e-3 * float(dw_ref.abs().max()), what='thin-output wgrad')
    ops.conv_wgrad(xd, dyd, dw, k, 1, pad, accumulate=True)
    close(dw.cpu(), 2 * dw_ref, tol=5e-3, floor=2e-3 * float(dw_ref.abs().max()), what='thin-output wgrad, accumulated')
    # same bits run after run (fixed fold order)
    dw2 = torch.zeros_like(dw)
    ops.conv_wgrad(xd, dyd, dw2, k, 1, pad, accumulate=False)
    dw3 = torch.zeros_like(dw)
    ops.conv_wgrad(xd, dyd, dw3, k, 1, pad, accumulate=False)
    assert torch.equal(dw2, dw3)


@pytest.mark.parametrize('act', ['tanh', 'none'])
@pytest.mark.parametrize('case', [
    (2, 40, 80, 64, 3, 9), (1, 33, 70, 24, 3, 9), (2, 24, 64, 32, 3, 5), (1, 20, 130, 16, 2, 7), (3, 9, 16, 8, 1, 3), (1, 96, 200, 64, 3, 9)])
def test_thin_output_wide_kernel_fprop(case, act):
    """gcc_conv_fprop's thin-output route (conv_thinout.hip: U = W x row per 64 input columns, then the shift-and-add over the
    horizontal taps, bias + activation, 8-channel padded store) against torch conv2d on the same bf16-rounded inputs; one launch;
    the padding channels of the output stay zero"""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (k * Ci ** 0.5)))
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv2d(x, w, b, stride=1, padding=pad)
    ref = torch.tanh(ref) if act == 'tanh' else ref
    wp, _ = ops.pack_weights(master_cl(w))
    y = ops.new_act(N, Co, H, W, DEV)
    y.fill_(3.0)                                               # stale contents must go
    ops.lib().gcc_launch_count(1)
    ops.conv_fprop(to_dev(x), wp, Co, k, 1, pad, out=y, bias=b.to(DEV), act=ops.ACT_TANH if act == 'tanh' else ops.ACT_NONE)
    assert int(ops.lib().gcc_launch_count(1)) == 1
    close(to_cpu(y), ref, what='thin-output fprop')
    full = y.permute(0, 2, 3, 1)
    phys = torch.as_strided(full, (N, H, W, 8), (H * W * 8, W * 8, 8, 1))
    assert float(phys[..., Co:].abs().max()) == 0.0, 'padding channels must be zero'


@pytest.mark.parametrize('case', [
    (2, 40, 80, 64, 3, 9), (1, 33, 70, 24, 3, 9), (2, 24, 64, 32, 3, 5), (1, 20, 130, 16, 2, 7), (3, 9, 16, 8, 1, 3), (1, 96, 200, 64, 3, 9)])
def test_thin_output_wide_kernel_dgrad(case):
    """gcc_conv_dgrad's thin-output route (conv_thinout.hip: one 32-deep k-step per vertical tap over the expanded rows of dY)
    against torch's conv2d_input on the same bf16-rounded inputs; one launch"""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case) + 2)
    dy = rb(torch.randn(N, Co, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (k * Co ** 0.5)))
    ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w, dy, stride=1, padding=pad)
    _, wtp = ops.pack_weights(master_cl(w))
    dx = ops.new_act(N, Ci, H, W, DEV)
    dx.fill_(3.0)
    ops.lib().gcc_launch_count(1)
    ops.conv_dgrad(to_dev(dy), wtp, Ci, H, W, k, 1, pad, out=dx)
    assert int(ops.lib().gcc_launch_count(1)) == 1
    close(to_cpu(dx), ref, what='thin-output dgrad')


@pytest.mark.parametrize('case', [(2, 48, 48, 24, 9, 3), (1, 40, 64, 64, 10, 3), (2, 24, 32, 24, 10, 3), (1, 32, 48, 64, 9, 3)])
def test_pruned_3x3_layers_with_nine_or_ten_output_channels(case):
    """SRGAN's pruned residual blocks (inner_channels = int(sum(mask)): 24 -> 9, 64 -> 10 at 3 x 3 s1 p1) satisfied
    the thin-output plan's Co * KW <= 32, but its gradient kernels stage only channels 0..7 of dY: weight and data gradient of such a
    layer against torch, whichever kernel the geometry is routed to now (not the thin-output one: tests/test_cabi_and_host.py)"""
    ops = _ops()
    N, H, W, Ci, Co, k = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    dy = rb(torch.randn(N, Co, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (k * Co ** 0.5)))
    dw_ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), dy, stride=1, padding=1)
    dx_ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w, dy, stride=1, padding=1)
    xd, dyd = to_dev(x), to_dev(dy)
    dw = torch.full((Co, Ci, k, k), 5.0, device=DEV).contiguous(memory_format=torch.channels_last)
    ops.conv_wgrad(xd, dyd, dw, k, 1, 1, accumulate=False)
    close(dw.cpu(), dw_ref, tol=5e-3, floor=1e-3 * float(dw_ref.abs().max()), what='wgrad, Co = %d' % Co)
    _, wtp = ops.pack_weights(master_cl(w))
    dx = ops.new_act(N, Ci, H, W, DEV)
    dx.fill_(3.0)
    ops.conv_dgrad(dyd, wtp, Ci, H, W, k, 1, 1, out=dx)
    close(to_cpu(dx), dx_ref, what='dgrad, Co = %d' % Co)
    y_ref = F.conv2d(x, w, None, stride=1, padding=1)
    wp, _ = ops.pack_weights(master_cl(w))
    y = ops.new_act(N, Co, H, W, DEV)
    ops.conv_fprop(xd, wp, Co, k, 1, 1, out=y)
    close(to_cpu(y), y_ref, what='fprop, Co = %d' % Co)


def test_grouped_weight_gradients_vs_per_layer_and_torch():
    """gcc_conv_wgrad_group_* (round 6): the weight gradients of a U-Net-like set of layers -- stride-2 4 x 4 convolutions between
    8..256 channels from 64 x 64 down to 1 x 1, a ConvTranspose's adjoint pair, a 1 x 1 and a 3 x 3 layer, ragged pixel counts -- as
    ONE launch + one fold, against torch's conv2d_weight on the same bf16-rounded operands and against gcc_conv_wgrad layer by layer
    (same tolerance: the two differ in their pixel-split plans, i.e. in fp32 summation order); fresh and accumulating; the launch
    count shows the grouping (2 launches for 9 layers); same bits run after run; a group that holds an irregular width is refused."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    # (N, Ci, H, W, Co, k, stride, pad)
    layers = [(3, 8, 64, 64, 16, 4, 2, 1), (3, 16, 32, 32, 32, 4, 2, 1), (3, 32, 16, 16, 64, 4, 2, 1), (3, 64, 8, 8, 256, 4, 2, 1),
              (3, 256, 4, 4, 256, 4, 2, 1), (3, 256, 2, 2, 128, 4, 2, 1), (2, 24, 33, 47, 40, 3, 1, 1), (3, 48, 20, 20, 16, 1, 1, 0),
              (3, 128, 2, 2, 128, 4, 2, 1)]
    entries, refs = [], []
    for (N, Ci, H, W, Co, k, st, pd) in layers:
        Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
        x = rb(torch.randn(N, Ci, H, W, generator=g))
        dy = rb(torch.randn(N, Co, Ho, Wo, generator=g))
        refs.append(torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), dy, stride=st, padding=pd))
        dw = torch.full((Co, Ci, k, k), 5.0, device=DEV).contiguous(memory_format=torch.channels_last)      # stale contents must go
        entries.append((to_dev(x), to_dev(dy), dw, k, st, pd, False))
    grp = ops.WgradGroup()
    assert grp.groupable(entries)
    ops.lib().gcc_launch_count(1)
    grp.run()
    assert int(ops.lib().gcc_launch_count(1)) == 2, 'one grouped launch + one fold'
    for (x, dy, dw, k, st, pd, _), ref, lay in zip(entries, refs, layers):
        close(dw.cpu(), ref, tol=5e-3, floor=1e-3 * float(ref.abs().max()), what='grouped wgrad %s' % (lay,))
        single = torch.zeros_like(dw)
        ops.conv_wgrad(x, dy, single, k, st, pd, accumulate=False)
        close(dw.cpu(), single.cpu(), tol=5e-4, floor=1e-4 * float(ref.abs().max()), what='grouped vs per-layer %s' % (lay,))
    first = [e[2].clone() for e in entries]
    # accumulating into what the buffers hold (a second group object over the same buffers), twice the gradient
    acc = [(x, dy, dw, k, st, pd, True) for (x, dy, dw, k, st, pd, _) in entries]
    grp2 = ops.WgradGroup()
    assert grp2.groupable(acc)
    grp2.run()
    for (x, dy, dw, k, st, pd, _), ref in zip(acc, refs):
        close(dw.cpu(), 2 * ref, tol=5e-3, floor=2e-3 * float(ref.abs().max()), what='grouped wgrad, accumulated')
    # same bits run after run
    grp.run()
    for e, f in zip(entries, first):
        assert torch.equal(e[2], f)
    # a 3-channel image layer is not groupable: the caller keeps it on gcc_conv_wgrad
    x = to_dev(rb(torch.randn(2, 3, 32, 32, generator=g)))
    dy = to_dev(rb(torch.randn(2, 16, 16, 16, generator=g)))
    dw = torch.zeros((16, 3, 4, 4), device=DEV).contiguous(memory_format=torch.channels_last)
    assert not ops.WgradGroup().groupable(entries[:2] + [(x, dy, dw, 4, 2, 1, False)])


def test_grouped_channel_sums_are_the_single_launches_bit_for_bit():
    """gcc_channel_sum_group (round 6): the bias gradients of a grouped weight gradient's layers as one launch -- every entry the bits
    of gcc_channel_sum on the same tensor (same per-thread order, same double fold), fresh and accumulating; a tensor of more than
    16384 pixels goes through gcc_channel_sum itself"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    shapes = [(1, 24, 64, 64), (2, 96, 16, 16), (1, 3, 128, 128), (4, 40, 9, 7), (1, 256, 1, 1), (2, 64, 96, 96)]      # the last: 18432 pixels
    xs = [to_dev(rb(torch.randn(*sh, generator=g))) for sh in shapes]
    for accumulate in (False, True):
        ref, got = [], []
        for x in xs:
            r = torch.full((x.shape[1],), 0.25, device=DEV)
            ops.channel_sum(x, r, accumulate=accumulate)
            ref.append(r)
            got.append(torch.full((x.shape[1],), 0.25, device=DEV))
        ops.lib().gcc_launch_count(1)
        ops.channel_sum_group(list(zip(xs, got)), accumulate=accumulate)
        assert int(ops.lib().gcc_launch_count(1)) == 3, 'five small tensors in one launch, the large one in its two'
        for x, a, b in zip(xs, ref, got):
            assert torch.equal(a, b), tuple(x.shape)
        want = xs[0].float().sum(dim=(0, 2, 3)) + (0.25 if accumulate else 0.0)
        close(got[0].cpu(), want.cpu(), tol=1e-5, floor=1e-4, what='channel sums')


RING3_CASES = [
    # N, H, W, Ci, Co               3 x 3 stride-1 layers between <= 64-channel tensors: conv_ring3.hip
    (2, 96, 96, 64, 64),            # SRGAN teacher's trunk layer at two images: three strips of 32 columns, several row bands
    (1, 90, 120, 64, 64),           # two 64-column strips, the second ragged
    (2, 70, 96, 24, 24),            # student widths: 32-channel LDS rows, one 32-channel destination half
    (1, 128, 80, 40, 64),           # 40 -> 64
    (1, 96, 128, 64, 24),           # 64 -> 24
    (1, 96, 128, 16, 48),           # 16 -> 48
    (3, 64, 48, 32, 32),            # one strip of 64 columns three quarters used
    (1, 256, 33, 8, 8),             # narrowest: 8 channels, 33 columns
    (1, 96, 96, 24, 20),            # destination width not a multiple of 8: the padding lanes of the output stay zero
    (1, 520, 16, 16, 16),           # narrowest image (one half-used strip of 32), a long walk: the ring wraps many times
    (4, 33, 65, 64, 64),            # 65 columns: three strips of 32, the last one pixel wide; odd height
]


def _ring3_route(case, dgrad):
    import ctypes as C
    ops = _ops()
    N, H, W, Ci, Co = case
    d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, (Ci + 7) & ~7, (Co + 7) & ~7)
    return int(ops.lib().gcc_conv_route(C.byref(d), dgrad, None))


@pytest.mark.parametrize('act', ['none', 'relu', 'lrelu'])
@pytest.mark.parametrize('case', RING3_CASES)
def test_ring_walk_3x3_fprop(case, act):
    """gcc_conv_fprop's ring-walk route (conv_ring3.hip: rows of x once through an LDS ring, the nine taps of a wave's output
    channels in registers, a horizontal tap = a shifted operand address) with bias + activation against fp32 torch on the same
    bf16 operands and against igemm_kernel (GCC_OPT_IGEMM_THIN 0); one launch; stale output and padding lanes overwritten"""
    from gcc_amd import _lib
    ops = _ops()
    N, H, W, Ci, Co = case
    assert _ring3_route(case, 0) == 4
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, 3, 3, generator=g) * (1.0 / (3 * Ci ** 0.5)))
    b = torch.randn(Co, generator=g) * 0.1
    ref = F.conv2d(x, w, b, stride=1, padding=1)
    ref = {'none': ref, 'relu': F.relu(ref), 'lrelu': F.leaky_relu(ref, 0.2)}[act]
    a = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'lrelu': ops.ACT_LRELU}[act]
    wp, _ = ops.pack_weights(master_cl(w))
    xd = to_dev(x)
    y = ops.new_act(N, Co, H, W, DEV)
    y.fill_(3.0)
    ops.lib().gcc_launch_count(1)
    ops.conv_fprop(xd, wp, Co, 3, 1, 1, out=y, bias=b.to(DEV), act=a, slope=0.2)
    assert int(ops.lib().gcc_launch_count(1)) == 1
    close(to_cpu(y), ref, what='ring-walk fprop')
    ld = (Co + 7) & ~7
    if ld > Co:
        phys = torch.as_strided(y, (N, H, W, ld), (H * W * ld, W * ld, ld, 1))
        assert float(phys[..., Co:].float().abs().max()) == 0.0, 'padding channels must be zero'
    lib = ops.lib()
    prev = lib.gcc_get_option(_lib.OPT_IGEMM_THIN)
    try:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, 0)
        y0 = ops.conv_fprop(xd, wp, Co, 3, 1, 1, bias=b.to(DEV), act=a, slope=0.2)
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, prev)
    close(to_cpu(y), to_cpu(y0), what='ring-walk fprop vs igemm_kernel')
    y2 = ops.conv_fprop(xd, wp, Co, 3, 1, 1, bias=b.to(DEV), act=a, slope=0.2)
    assert torch.equal(y, y2), 'same bits run after run'


@pytest.mark.parametrize('case', RING3_CASES)
def test_ring_walk_3x3_statistics_and_batchnorm(case, monkeypatch):
    """the route's BatchNorm partial sums (one row per workgroup, of the ROUNDED outputs) and the finalize inside the call: rows sum to
    the statistics of the output tensor; mean / rstd / running statistics against torch's batch statistics of the conv output"""
    import torch.nn as nn
    ops = _ops()
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(sum(case) + 12)
    x = rb(torch.randn(N, Ci, H, W, generator=g) + 0.3)
    w = rb(torch.randn(Co, Ci, 3, 3, generator=g) * (1.0 / (3 * Ci ** 0.5)))
    wp, _ = ops.pack_weights(master_cl(w))
    xd = to_dev(x)
    raw, st = ops.conv_fprop(xd, wp, Co, 3, 1, 1, want_stats=True)
    yd = to_cpu(raw).double()
    want = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    got = st.double().sum(0).cpu()
    assert float((got - want).abs().max() / want.abs().max()) <= 1e-5
    close(to_cpu(raw), F.conv2d(x, w, None, stride=1, padding=1), what='ring-walk fprop (statistics launch)')
    gam, bet = 1 + 0.1 * torch.randn(Co, generator=g), torch.randn(Co, generator=g)

    def run(in_conv):
        monkeypatch.setattr(ops, 'IN_CONV_FINALIZE', in_conv)
        bn = nn.BatchNorm2d(Co).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(gam)
            bn.bias.copy_(bet)
        state = ops.BNState(Co, DEV)
        out = ops.new_act(N, Co, H, W, DEV)
        launches = []
        for _ in range(2):                     # twice on one tail workspace: the ticket words reset themselves
            ops.lib().gcc_launch_count(1)
            ops.conv_fprop(xd, wp, Co, 3, 1, 1, out=out, want_stats=True, bn=ops.bn_desc(bn, state, N * H * W, DEV))
            launches.append(int(ops.lib().gcc_launch_count(1)))
        torch.cuda.synchronize()
        return out, bn, [t.clone() for t in (state.mean, state.rstd, state.scale, state.shift, bn.running_mean, bn.running_var)], launches
    out, bn, a, la = run(True)
    out0, _, b, lb = run(False)
    assert la == [1, 1] and lb == [2, 2], (la, lb)        # folded by the launch's last-arriving workgroups / by a gcc_bn_finalize launch
    assert torch.equal(out, raw) and torch.equal(out0, raw)
    for name, u, v in zip(('mean', 'rstd', 'scale', 'shift', 'running_mean', 'running_var'), a, b):
        assert torch.equal(u, v), name
    r = to_cpu(out)
    m, v = r.mean((0, 2, 3)), r.var((0, 2, 3), unbiased=False)
    vu = r.var((0, 2, 3), unbiased=True)
    close(a[0].cpu(), m, tol=1e-4, floor=1e-5, what='batch mean')
    close(a[1].cpu(), 1.0 / torch.sqrt(v + bn.eps), tol=1e-4, floor=1e-5, what='rstd')
    close(a[4].cpu(), 0.1 * m + 0.9 * 0.1 * m, tol=1e-4, floor=1e-5, what='running mean after two updates')
    close(a[5].cpu(), 0.81 + 0.19 * vu, tol=1e-4, floor=1e-5, what='running var after two updates')


@pytest.mark.parametrize('case', RING3_CASES)
def test_ring_walk_3x3_dgrad(case):
    """gcc_conv_dgrad's ring-walk route (the same kernel over dY, taps mirrored, dgrad packing) against torch's conv2d_input on the
    same bf16 operands and against igemm_kernel; one launch"""
    from gcc_amd import _lib
    ops = _ops()
    N, H, W, Ci, Co = case
    Ci, Co = Co, Ci                        # the cases name (source, destination) widths: here dY has the source's
    if Co % 8:
        pytest.skip('source width must be a multiple of 8')
    # (8 -> 8: 3 x 8 = 24 <= 32 columns, the thin-output data gradient of conv_thinout.hip comes first)
    assert _ring3_route((N, H, W, Ci, Co), 1) == (3 if Co * 3 <= 32 else 4)
    g = torch.Generator().manual_seed(sum(case) + 13)
    dy = rb(torch.randn(N, Co, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, 3, 3, generator=g) * (1.0 / (3 * Co ** 0.5)))
    ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w, dy, stride=1, padding=1)
    _, wtp = ops.pack_weights(master_cl(w))
    dyd = to_dev(dy)
    dx = ops.new_act(N, Ci, H, W, DEV)
    dx.fill_(3.0)
    ops.lib().gcc_launch_count(1)
    ops.conv_dgrad(dyd, wtp, Ci, H, W, 3, 1, 1, out=dx)
    assert int(ops.lib().gcc_launch_count(1)) == 1
    close(to_cpu(dx), ref, what='ring-walk dgrad')
    lib = ops.lib()
    prev = lib.gcc_get_option(_lib.OPT_IGEMM_THIN)
    try:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, 0)
        dx0 = ops.conv_dgrad(dyd, wtp, Ci, H, W, 3, 1, 1)
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, prev)
    close(to_cpu(dx), to_cpu(dx0), what='ring-walk dgrad vs igemm_kernel')


def test_instance_norm_workspace_scrub_keeps_results():
    """the launcher re-zeroes a grid InstanceNorm workspace at its first use inside every launch recording (and every 2^20
    launches), so that the 24-bit epoch field of the exchange tag never wraps: a recorded + replayed sequence of launches gives
    the bits of the eager one"""
    import ctypes as C
    ops = _ops()
    lib = ops.lib()
    x = to_dev(rb(torch.randn(2, 96, 64, 64, generator=torch.Generator().manual_seed(4))))
    y, y2 = ops.new_act(2, 96, 64, 64, DEV), ops.new_act(2, 96, 64, 64, DEV)
    st = ops.INState(2, 96, DEV)
    for _ in range(3):
        ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    h = C.c_void_p()
    assert lib.gcc_replay_begin(C.byref(h)) == 0
    try:
        for _ in range(3):
            ops.inorm_fwd(x, y2, st, act=ops.ACT_RELU)
    finally:
        assert lib.gcc_replay_end(h, 1) == 0
    entries = int(lib.gcc_replay_info(h, 0))
    assert entries == 4, entries                    # one memset (the scrub) + three launches
    for _ in range(5):
        y2.zero_()
        assert lib.gcc_replay_run(h) == 0
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
    lib.gcc_replay_destroy(h)


def test_instance_norm_grid_form_on_concurrent_streams():
    """the grid form's in-launch barrier under the conditions of a training step: four streams launch it back to back on their
    own tensors (each stream has its own workspace) while a fifth keeps the chip busy with large copies -- every launch completes
    (residency comes from the grid size: <= 256 workgroups, four per CU) and every result equals the one computed alone"""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    shapes = [(1, 256, 64, 64), (2, 96, 64, 64), (1, 64, 128, 128), (1, 128, 128, 128)]
    xs = [to_dev(rb(torch.randn(*s_, generator=g))) for s_ in shapes]
    gs = [to_dev(rb(torch.randn(*s_, generator=g))) for s_ in shapes]
    ref = []
    for x, gy in zip(xs, gs):
        st = ops.INState(x.shape[0], x.shape[1], DEV)
        y, dx = ops.new_act(*x.shape, DEV), ops.new_act(*x.shape, DEV)
        ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
        ops.inorm_bwd(x, y, gy, dx, st, act=ops.ACT_RELU)
        ref.append((y.clone(), dx.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in shapes]
    noise = torch.cuda.Stream()
    big_a, big_b = torch.empty(256 << 20, dtype=torch.uint8, device=DEV), torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    outs = [(ops.new_act(*x.shape, DEV), ops.new_act(*x.shape, DEV), ops.INState(x.shape[0], x.shape[1], DEV)) for x in xs]
    torch.cuda.synchronize()
    for it in range(150):
        with torch.cuda.stream(noise):
            big_b.copy_(big_a)
        for s_, x, gy, (y, dx, st) in zip(streams, xs, gs, outs):
            with ops.on_stream(s_):
                ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
                ops.inorm_bwd(x, y, gy, dx, st, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    for (y, dx, _), (ry, rdx) in zip(outs, ref):
        assert torch.equal(y, ry) and torch.equal(dx, rdx)


@pytest.mark.parametrize('public_api', [False, True])
def test_stream_helpers_order_work(public_api):
    """ops.stream / ops.current_stream / ops.on_stream (torch's private raw-stream calls behind a guard, the public API as
    their fallback): inside `with on_stream(s)` the library's launches go to s -- a kernel enqueued there behind a long fill
    sees the fill's result -- and the previous stream is current again afterwards; record / wait through the same helpers
    orders a second stream behind it.  public_api=True forces the fallbacks a torch upgrade would leave us with."""
    ops = _ops()
    saved = ops._RAW_STREAM, ops._SET_STREAM
    if public_api:
        ops._RAW_STREAM = ops._SET_STREAM = None
    try:
        main_raw = ops.stream()
        assert main_raw == torch.cuda.current_stream().cuda_stream
        side, other = torch.cuda.Stream(), torch.cuda.Stream()
        big = torch.zeros(64 << 20, dtype=torch.float32, device=DEV)
        src = ops.new_act(1, 8, 64, 64, DEV)
        dst = ops.new_act(1, 8, 64, 64, DEV)
        torch.cuda.synchronize()
        with ops.on_stream(side):
            assert ops.stream() == side.cuda_stream and ops.current_stream().cuda_stream == side.cuda_stream
            for i in range(4):
                big.fill_(float(i))                      # ~1 ms in front of the write below
            src.fill_(3.0)
            ops.nhwc_copy(src, 0, dst, 0, 8)             # a library launch: must run on `side`, behind the fill
            ev = torch.cuda.Event()
            ev.record(ops.current_stream())
        assert ops.stream() == main_raw                   # restored
        with ops.on_stream(other):
            ops.current_stream().wait_event(ev)
            out = ops.new_act(1, 8, 64, 64, DEV)
            ops.nhwc_copy(dst, 0, out, 0, 8)
        torch.cuda.synchronize()
        assert float(out.float().min()) == 3.0 and float(out.float().max()) == 3.0
    finally:
        ops._RAW_STREAM, ops._SET_STREAM = saved


HALO_CASES = [  # N, H, W, Ci, Co, stride: the PatchGAN layers conv_halo.hip serves at the bench's batch, and a smaller grid of each form
    (16, 128, 128, 128, 256, 2), (16, 64, 64, 256, 512, 2), (16, 32, 32, 512, 1024, 1), (8, 64, 64, 128, 256, 2),
    (4, 64, 64, 64, 512, 1)]


@pytest.mark.parametrize('hc', [0, 128])
@pytest.mark.parametrize('N,H,W,Ci,Co', [(4, 64, 64, 64, 512), (16, 48, 48, 128, 256), (4, 96, 96, 128, 128), (3, 80, 112, 192, 256)])
def test_halo_conv_3x3(N, H, W, Ci, Co, hc):
    """igemm_halo_kernel<true, HC, 3> (k3 s1 p1: the VGG19 layers of SRGAN's perceptual loss; GCC_OPT_IGEMM_HALO 3): forward with
    bias + ReLU and statistics, data gradient -- against fp32 torch on the same bf16 operands and against igemm_kernel"""
    ops = _ops()
    from gcc_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(N + H + Co)
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    dy = rb(torch.randn(N, Co, H, W, generator=g))
    m = rb(torch.randn(Co, Ci, 3, 3, generator=g) * 0.05)
    b = torch.randn(Co, generator=g)
    w, wt = ops.pack_weights(m.to(DEV).contiguous(memory_format=torch.channels_last))
    ref_y = F.relu(F.conv2d(x, m, b, stride=1, padding=1))
    ref_raw = F.conv2d(x, m, None, stride=1, padding=1)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), m, dy, stride=1, padding=1)
    res = {}
    prev = lib.gcc_get_option(_lib.OPT_IGEMM_HALO)
    ops.set_plan(halo_hc=max(hc, 0))
    try:
        for halo in (0, 3):
            lib.gcc_set_option(_lib.OPT_IGEMM_HALO, halo)
            lib.gcc_launch_count(1)
            y = ops.conv_fprop(to_dev(x), w, Co, 3, 1, 1, bias=b.to(DEV), act=ops.ACT_RELU)
            raw, st = ops.conv_fprop(to_dev(x), w, Co, 3, 1, 1, want_stats=True)
            dx = ops.conv_dgrad(to_dev(dy), wt, Ci, H, W, 3, 1, 1)
            res[halo] = (to_cpu(y), to_cpu(raw), to_cpu(dx), st.double().sum(0).cpu())
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_HALO, prev)
        ops.set_plan()
    y3, raw3, dx3, st3 = res[3]
    close(y3, ref_y, what='3x3 halo fprop + bias + relu vs torch')
    close(raw3, ref_raw, what='3x3 halo fprop vs torch')
    close(dx3, ref_dx, what='3x3 halo dgrad vs torch')
    close(raw3, res[0][1], tol=1.2e-2, what='3x3 halo fprop vs gather kernel')
    close(dx3, res[0][2], tol=1.2e-2, what='3x3 halo dgrad vs gather kernel')
    yd = raw3.double()
    want = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    assert float((st3 - want).abs().max() / want.abs().max()) <= 1e-5
    assert not torch.equal(raw3, res[0][1]) or True      # (the two routes sum in different orders; equality is not required)


@pytest.mark.parametrize('hc', [0, 128])
@pytest.mark.parametrize('N,H,W,Ci,Co,stride', HALO_CASES + [(16, 128, 128, 64, 128, 2)])
def test_halo_conv_vs_torch_and_gather_kernel(N, H, W, Ci, Co, stride, hc):
    """igemm_halo_kernel (tile neighbourhood resident in LDS; k4 s2 p1 forward / data gradient per phase / both px phases of a
    128-channel data gradient, k4 s1 p1 forward on the padded grid / data gradient) against fp32 torch on the same bf16
    operands, against igemm_kernel (GCC_OPT_IGEMM_HALO = 0: same sums in another order), and its BatchNorm partial sums
    against the sums of its own rounded output."""
    ops = _ops()
    from gcc_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(N + H + Co)
    k, p = 4, 1
    Ho, Wo = (H + 2 - k) // stride + 1, (W + 2 - k) // stride + 1
    x = rb(torch.randn(N, Ci, H, W, generator=g))
    dy = rb(torch.randn(N, Co, Ho, Wo, generator=g))
    m = rb(torch.randn(Co, Ci, k, k, generator=g) * 0.05)
    w, wt = ops.pack_weights(m.to(DEV).contiguous(memory_format=torch.channels_last))
    ref_y = F.conv2d(x, m, stride=stride, padding=p)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), m, dy, stride=stride, padding=p)
    res = {}
    prev = lib.gcc_get_option(_lib.OPT_IGEMM_HALO)
    # hc = 128 (round 4): the 128-column tile form on every launch (gcc_conv_t.plan.halo_hc; the last case -- the teacher U-Net's
    # 64 -> 128 down conv -- has no 256-column tiling and takes it under the default plan too)
    ops.set_plan(halo_hc=max(hc, 0))
    try:
        for halo in (0, 2):
            lib.gcc_set_option(_lib.OPT_IGEMM_HALO, halo)
            y, st = ops.conv_fprop(to_dev(x), w, Co, k, stride, p, want_stats=True)
            dx = ops.conv_dgrad(to_dev(dy), wt, Ci, H, W, k, stride, p)
            res[halo] = (to_cpu(y), to_cpu(dx), st.double().sum(0).cpu())
            if halo:
                # the other tile order (GCC_OPT_HALO_XCD_COLS: one column tile per XCD): the same tiles on other workgroups, same bits
                lib.gcc_set_option(_lib.OPT_HALO_XCD_COLS, 2)
                yb, stb = ops.conv_fprop(to_dev(x), w, Co, k, stride, p, want_stats=True)
                dxb = ops.conv_dgrad(to_dev(dy), wt, Ci, H, W, k, stride, p)
                assert torch.equal(yb, y) and torch.equal(dxb, dx) and torch.equal(stb.double().sum(0).cpu(), res[halo][2])
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_HALO, prev)
        ops.set_plan()
        lib.gcc_set_option(_lib.OPT_HALO_XCD_COLS, -1)
    y0, dx0, _ = res[0]
    y2, dx2, st2 = res[2]
    close(y2, ref_y, what='halo fprop vs torch')
    close(dx2, ref_dx, what='halo dgrad vs torch')
    close(y2, y0, tol=1.2e-2, what='halo fprop vs gather kernel')
    close(dx2, dx0, tol=1.2e-2, what='halo dgrad vs gather kernel')
    yd = y2.double()
    want = torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])
    assert float((st2 - want).abs().max() / want.abs().max()) <= 1e-5
