"""CPU oracle for the GCC Pix2Pix hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a from-scratch, *functional* restatement (plain PyTorch-CPU fp32, parameters held in a
flat ``dict`` keyed by the reference's ``state_dict`` names) of what one iteration of the
reference's ``Pix2PixModel`` computes.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product (``gcc_amd``) never does and fails
loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_fixtures.py`` imports the real reference
(``/root/reference``, stub-imported as SURVEY.md Appendix B describes) in the authoring container
and stores its outputs (eval images, hooked features, every loss scalar, post-step weights / BN
running statistics / alpha, prune cfgs) under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.

Reference anchors (all relative to /root/reference):
  unet_forward ............ models/Pix2Pix.py:20-130   (UnetSkipConnectionBlock / UnetGenertor)
  patchgan_forward ........ models/Pix2Pix.py:267-348  (NLayerDiscriminator / MaskNLayerDiscriminator)
  gate_mask / gate ........ models/DifferentiableOp.py:22-59
  gan_loss ................ models/GANLoss.py:38-59
  gram .................... models/Pix2Pix.py:733-740
  Pix2PixOracle.optimize_parameters  models/Pix2Pix.py:464-583
  Pix2PixOracle.optimizer_netD_arch  models/Pix2Pix.py:479-511, 585-593
  adam_step ............... torch.optim.Adam as configured at models/Pix2Pix.py:382,415,430-431
  lr_lambda_linear ........ utils/util.py:288-303
  scale_prune_cfg / norm_prune_cfg / max_min_* ... models/Pix2Pix.py:754-902

The in-place aliasing of the reference (SURVEY.md hazard H1) is restated *explicitly*: the skip
concatenation and the hooked tensors are written as the post-activation values the reference ends
up reading, instead of relying on in-place mutation.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU = 0.2

# ----------------------------------------------------------------------------------------------
# optional storage-precision emulation: when EMULATE_BF16 is True every tensor the MI355X path keeps
# in bf16 (conv weights as the MFMA sees them, conv outputs, activated tensors, and their gradients
# on the way back) is rounded to bf16 here too.  Tests use it to MEASURE the deviation bf16 storage
# alone induces, so that tolerances are derived rather than guessed.  Off by default: the oracle
# proper is fp32 like the reference.
# ----------------------------------------------------------------------------------------------
EMULATE_BF16 = False


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _q(x):
    return _RoundBF16.apply(x) if EMULATE_BF16 else x


class _RoundFwdOnly(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _qw(w):
    return _RoundFwdOnly.apply(w) if (EMULATE_BF16 and w is not None and w.dim() == 4) else w


# ----------------------------------------------------------------------------------------------
# names
# ----------------------------------------------------------------------------------------------
def unet_block_prefix(d: int) -> str:
    """state_dict prefix of the U-Net block at depth d (0 = outermost). models/Pix2Pix.py:85-127."""
    if d == 0:
        return 'model'
    return 'model.model.1' + '.model.3' * (d - 1)


def unet_dropout_depths(num_downs: int) -> List[int]:
    """Depths whose up path ends in Dropout(0.5): the ``num_downs-5`` blocks built by the loop at
    models/Pix2Pix.py:95-102 (they sit directly above the innermost block)."""
    return [num_downs - 2 - i for i in range(num_downs - 5)]


def unet_hook_names(num_downs: int = 8) -> List[str]:
    """models/Pix2Pix.py:366-369"""
    return ['model.model.1.model.2',
            'model.model.1.model.3.model.3.model.2',
            'model.model.1.model.3.model.3.model.4',
            'model.model.1.model.4']


# ----------------------------------------------------------------------------------------------
# small ops
# ----------------------------------------------------------------------------------------------
def batch_norm(sd: SD, prefix: str, x: Tensor, train: bool) -> Tensor:
    """nn.BatchNorm2d(affine, track_running_stats), momentum .1, eps 1e-5.  In train mode the
    running statistics in ``sd`` are updated in place (unbiased variance), like the module."""
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if train:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            key = prefix + '.num_batches_tracked'
            if key in sd:
                sd[key] += 1
    else:
        mean, var = rm, rv
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xhat * w[None, :, None, None] + b[None, :, None, None]


def gate_mask(alpha: Tensor, threshold: float) -> Tensor:
    """(sign(alpha - tau) + 1) / 2  --  alpha == tau gives 0.5.  models/DifferentiableOp.py:25-26"""
    return (torch.sign(alpha - threshold) + 1) / 2


class _MaskSTE(torch.autograd.Function):
    """Straight-through: d mask / d alpha := 1.  models/DifferentiableOp.py:22-32"""

    @staticmethod
    def forward(ctx, alpha, threshold):
        return gate_mask(alpha, threshold)

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None


def gate(x: Tensor, alpha: Tensor, threshold: float) -> Tensor:
    """y = x * m[c]; dalpha[c] = sum_{n,h,w} dy*x (no mask factor), dx = dy*m."""
    m = _MaskSTE.apply(alpha, threshold)
    return x * m[None, :, None, None]


def gan_loss(mode: str, pred: Tensor, target_is_real: bool, for_discriminator: bool = True) -> Tensor:
    """models/GANLoss.py:38-59"""
    if mode == 'lsgan':
        t = torch.ones_like(pred) if target_is_real else torch.zeros_like(pred)
        return F.mse_loss(pred, t)
    if mode == 'vanilla':
        t = torch.ones_like(pred) if target_is_real else torch.zeros_like(pred)
        return F.binary_cross_entropy_with_logits(pred, t)
    if mode == 'wgangp':
        return -pred.mean() if target_is_real else pred.mean()
    if mode == 'hinge':
        if for_discriminator:
            z = (pred - 1) if target_is_real else (-pred - 1)
            return -torch.min(z, torch.zeros_like(z)).mean()     # binary min: ties get half the gradient
        assert target_is_real
        return -pred.mean()
    raise NotImplementedError('gan mode %s not implemented' % mode)


def gram(x: Tensor) -> Tensor:
    """G = F F^T / (c h w), F = x.view(b, c, hw).  models/Pix2Pix.py:733-740"""
    b, c, h, w = x.shape
    f = x.reshape(b, c, h * w)
    return torch.bmm(f, f.transpose(1, 2)) / (c * h * w)


def rmse(a: Tensor, b: Tensor) -> Tensor:
    return torch.sqrt(F.mse_loss(a, b))


# ----------------------------------------------------------------------------------------------
# U-Net generator
# ----------------------------------------------------------------------------------------------
def unet_forward(sd: SD, x: Tensor, num_downs: int = 8, train: bool = True,
                 dropout: bool = False, dropout_masks: Optional[Dict[int, Tensor]] = None,
                 features: Optional[OrderedDict] = None, drop_depths: Optional[List[int]] = None) -> Tensor:
    """8-down / 8-up U-Net with forced BatchNorm (models/Pix2Pix.py:26), written depth by depth.

    e[d]  : pre-activation output of the down conv (+BN for 0<d<D-1) at depth d
    u     : output of the up conv + BN at a depth
    skip  : because the reference's LeakyReLU is in-place, the tensor concatenated at depth d is
            leaky_relu(e[d-1]), and the tensor a hook on the down-BN sees is leaky_relu(e[d]).
    ``features`` (if given) receives the four hooked tensors under the reference's module names.
    ``dropout_masks[d]`` optionally injects the (already 1/(1-p)-scaled) mask for depth d.
    """
    # A pruned generator may have lost inner blocks (models/Pix2Pix.py:87, 97): the state_dict says how many are built (nesting
    # goes by position) and of what kind the last one is -- the innermost kind (conv, ReLU, convT, BN) or a loop block around
    # Identity (:59-67: conv, BN, ReLU, convT, BN).  ``drop_depths`` then names the positions of the loop blocks that are left.
    D = num_downs
    while D > 1 and (unet_block_prefix(D - 1) + '.model.1.weight') not in sd:
        D -= 1
    ident = (unet_block_prefix(D - 1) + '.model.5.weight') in sd
    hook = unet_hook_names(D)
    if drop_depths is None:
        drop_depths = unet_dropout_depths(num_downs) if D == num_downs else [d for d in range(4, D - (0 if ident else 1))]
    drop_depths = drop_depths if dropout else []

    def conv(t, key):
        return F.conv2d(t, _qw(sd[key + '.weight']), sd.get(key + '.bias'), stride=2, padding=1)

    def convT(t, key):
        return F.conv_transpose2d(t, _qw(sd[key + '.weight']), sd.get(key + '.bias'), stride=2, padding=1)

    e: List[Tensor] = [None] * D
    # (bf16 emulation: the HIP path never stores e[0] -- the outermost conv's launch writes leaky_relu(e0) and relu(e0) straight
    # from its fp32 accumulators, gcc_epilogue_t.y2 -- so the only roundings are those of the two activated copies below)
    e[0] = conv(_q(x), 'model.model.0')
    for d in range(1, D):
        p = unet_block_prefix(d)
        z = _q(conv(_q(F.leaky_relu(e[d - 1], LRELU)), p + '.model.1'))
        if d < D - 1 or ident:
            z = batch_norm(sd, p + '.model.2', z, train)
            if features is not None and (p + '.model.2') in hook:
                # hazard H1: the hooked BatchNorm output is overwritten in place by the next block's LeakyReLU -- or, when the
                # block wraps Identity, by its own ReLU
                features[p + '.model.2'] = _q(F.leaky_relu(z, LRELU)) if d < D - 1 else _q(F.relu(z))
        e[d] = z

    p = unet_block_prefix(D - 1)
    if not ident:
        u = batch_norm(sd, p + '.model.4', _q(convT(_q(F.relu(e[D - 1])), p + '.model.3')), train)
    else:
        r = _q(F.relu(e[D - 1]))
        if features is not None and (p + '.model.4') in hook:
            features[p + '.model.4'] = r
        u = batch_norm(sd, p + '.model.6', _q(convT(r, p + '.model.5')), train)
        if (D - 1) in drop_depths and train:
            u = u * dropout_masks[D - 1] if (dropout_masks is not None and (D - 1) in dropout_masks) else F.dropout(u, 0.5, True)
    cat = torch.cat([F.leaky_relu(e[D - 2], LRELU), u], 1)
    for d in range(D - 2, 0, -1):
        p = unet_block_prefix(d)
        r = _q(F.relu(cat))
        if features is not None and (p + '.model.4') in hook:
            features[p + '.model.4'] = r
        u = batch_norm(sd, p + '.model.6', _q(convT(r, p + '.model.5')), train)
        if d in drop_depths and train:
            if dropout_masks is not None and d in dropout_masks:
                u = u * dropout_masks[d]
            else:
                u = F.dropout(u, 0.5, True)
        cat = torch.cat([F.leaky_relu(e[d - 1], LRELU), u], 1)
    out = _q(torch.tanh(convT(_q(F.relu(cat)), 'model.model.3')))
    if features is not None:       # keep the reference's hook-firing order
        for k in hook:
            if k in features:
                features.move_to_end(k)
    return out


# ----------------------------------------------------------------------------------------------
# MobileResnet generator (models/Pix2Pix.py:132-265, models/CycleGAN.py:77-138)
# ----------------------------------------------------------------------------------------------
def resnet_layout(sd: SD):
    """(stem conv indices, block indices, up-conv indices, last conv index) from the state_dict keys: the
    Sequential is [pad, conv7, IN, ReLU, conv3s2, IN, ReLU, conv3s2, IN, ReLU, blocks..., convT, IN, ReLU, convT,
    IN, ReLU, pad, conv7, tanh]; removed blocks (cfg entry 0) shift the later indices."""
    tops = sorted({int(k.split('.')[1]) for k in sd})
    blocks = sorted({int(k.split('.')[1]) for k in sd if '.conv_block.' in k})
    rest = [i for i in tops if i not in blocks]
    return rest[:3], blocks, rest[3:5], rest[5]


def instance_norm(x: Tensor) -> Tensor:
    return F.instance_norm(x, eps=1e-5)


def mobile_resnet_forward(sd: SD, x: Tensor, features: Optional[OrderedDict] = None,
                          hook_idx: Sequence[int] = (9, 12, 15, 18)) -> Tensor:
    """InstanceNorm2d(affine=False) everywhere, every conv has a bias (use_bias follows the norm type), reflect
    padding in front of the 7x7 convs and inside the separable convs, residual blocks
    x + IN(pw(IN(dw(pad(relu(IN(pw(IN(dw(pad(x))))))))))), dropout rate 0."""
    stem, blocks, ups, last = resnet_layout(sd)

    def W(i, sfx=''):
        return _qw(sd['model.%d%s.weight' % (i, sfx)]), sd.get('model.%d%s.bias' % (i, sfx))

    w, b = W(stem[0])
    h = _q(F.relu(instance_norm(_q(F.conv2d(F.pad(_q(x), (3,) * 4, mode='reflect'), w, b)))))
    for i in stem[1:]:
        w, b = W(i)
        h = _q(F.relu(instance_norm(_q(F.conv2d(h, w, b, stride=2, padding=1)))))
    if features is not None and (stem[2] + 2) in hook_idx:
        features['model.%d' % (stem[2] + 2)] = h

    def sep(t, i, j):
        wd, bd = sd['model.%d.conv_block.%d.conv.0.weight' % (i, j)], sd['model.%d.conv_block.%d.conv.0.bias' % (i, j)]
        wp, bp = _qw(sd['model.%d.conv_block.%d.conv.2.weight' % (i, j)]), sd['model.%d.conv_block.%d.conv.2.bias' % (i, j)]
        t = _q(F.conv2d(F.pad(t, (1,) * 4, mode='reflect'), wd, bd, groups=t.shape[1]))
        t = _q(instance_norm(t))
        return _q(F.conv2d(t, wp, bp))
    for i in blocks:
        t = _q(F.relu(instance_norm(sep(h, i, 1))))
        t = instance_norm(sep(t, i, 6))
        h = _q(h + t)
        if features is not None and i in hook_idx:
            features['model.%d' % i] = h
    for i in ups:
        w, b = W(i)
        h = _q(F.relu(instance_norm(_q(F.conv_transpose2d(h, w, b, stride=2, padding=1, output_padding=1)))))
    w, b = W(last)
    return _q(torch.tanh(F.conv2d(F.pad(h, (3,) * 4, mode='reflect'), w, b)))


def mobile_resnet_shapes(ngf: int, cfg: Optional[Sequence[int]] = None, n_blocks: int = 9, in_nc=3, out_nc=3):
    """state_dict shapes of MobileResnetGenerator (cfg: 23 ints; None = unpruned)"""
    if cfg is None:
        cfg = [ngf, 2 * ngf, 4 * ngf] + [4 * ngf] * (2 * n_blocks) + [2 * ngf, ngf]
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def conv(i, co, ci, k, sfx=''):
        shp['model.%d%s.weight' % (i, sfx)] = (co, ci, k, k)
        shp['model.%d%s.bias' % (i, sfx)] = (co,)
    conv(1, cfg[0], in_nc, 7)
    conv(4, cfg[1], cfg[0], 3)
    conv(7, cfg[2], cfg[1], 3)
    idx, ci = 10, 2
    for b in range(n_blocks):
        c_in, c_mid, c_out = cfg[ci], cfg[ci + 1], cfg[ci + 2]
        ci += 2
        if c_mid == 0:
            continue
        for j, (a, o) in ((1, (c_in, c_mid)), (6, (c_mid, c_out))):
            shp['model.%d.conv_block.%d.conv.0.weight' % (idx, j)] = (a, 1, 3, 3)
            shp['model.%d.conv_block.%d.conv.0.bias' % (idx, j)] = (a,)
            shp['model.%d.conv_block.%d.conv.2.weight' % (idx, j)] = (o, a, 1, 1)
            shp['model.%d.conv_block.%d.conv.2.bias' % (idx, j)] = (o,)
        idx += 1
    for u in range(2):
        shp['model.%d.weight' % idx] = (cfg[ci], cfg[ci + 1], 3, 3)       # ConvTranspose2d [in, out, k, k]
        shp['model.%d.bias' % idx] = (cfg[ci + 1],)
        ci += 1
        idx += 3
    conv(idx + 1, out_nc, cfg[ci], 7)
    return shp


# ----------------------------------------------------------------------------------------------
# PatchGAN discriminators
# ----------------------------------------------------------------------------------------------
def patchgan_layout(masked: bool, n_layers: int = 3):
    """Sequential indices of (conv, bn, gate) per layer.  models/Pix2Pix.py:280-300 / 320-343"""
    L = []
    if not masked:
        L.append((0, None, None))
        i = 2
        for _ in range(n_layers):
            L.append((i, i + 1, None))
            i += 3
        L.append((i, None, None))
    else:
        L.append((0, None, 2))
        i = 3
        for _ in range(n_layers):
            L.append((i, i + 1, i + 2))
            i += 4
        L.append((i, None, None))
    return L


def patchgan_forward(sd: SD, x: Tensor, masked: bool = False, threshold: float = 0.5,
                     train: bool = True, features: Optional[OrderedDict] = None,
                     hook_names: Sequence[str] = ()) -> Tensor:
    """conv(k4 s2)+LReLU [+gate]; 2x conv(k4 s2)+BN[+gate]+LReLU; conv(k4 s1)+BN[+gate]+LReLU;
    conv(k4 s1)->1.  Plain D: a hook on a BN sees the post-LeakyReLU tensor (in-place, H1).
    Masked D: the gate multiplies out-of-place, so a hook on a BN sees the raw BN output."""
    lay = patchgan_layout(masked)
    h = x
    n = len(lay)
    for li, (ci, bi, gi) in enumerate(lay):
        stride = 2 if li < n - 2 else 1
        h = _q(F.conv2d(_q(h), _qw(sd['model.%d.weight' % ci]), sd.get('model.%d.bias' % ci), stride=stride, padding=1))
        if li == n - 1:
            break
        if bi is not None:
            # CycleGAN's plain (teacher) D uses InstanceNorm2d(affine=False) and conv biases: models/CycleGAN.py:139-177
            h = batch_norm(sd, 'model.%d' % bi, h, train) if ('model.%d.weight' % bi) in sd else _q(instance_norm(h))
            raw_bn = h
        if li == 0:
            h = F.leaky_relu(h, LRELU)
            if gi is not None:
                h = gate(h, sd['model.%d.alpha' % gi], threshold)
        else:
            if gi is not None:
                h = gate(h, sd['model.%d.alpha' % gi], threshold)
            h = F.leaky_relu(h, LRELU)
        if features is not None and bi is not None and ('model.%d' % bi) in hook_names:
            features['model.%d' % bi] = raw_bn if masked else h
    return h


# ----------------------------------------------------------------------------------------------
# init / optimizer / schedule
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def adam_step(params: List[Tensor], grads: List[Tensor], state: dict, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.Adam (no weight decay, no amsgrad): one step, in place on ``params``."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    b1, b2 = betas
    if 'm' not in state:
        state['m'] = [torch.zeros_like(p) for p in params]
        state['v'] = [torch.zeros_like(p) for p in params]
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    for p, g, m, v in zip(params, grads, state['m'], state['v']):
        if g is None:
            continue
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def lr_lambda_linear(epoch: int, epoch_count: int, n_epochs: int, n_epochs_decay: int) -> float:
    """utils/util.py:291-293"""
    return 1.0 - max(0, epoch + epoch_count - n_epochs) / float(n_epochs_decay + 1)


def init_state_dict(shapes: Dict[str, Tuple[int, ...]], gen: torch.Generator) -> SD:
    """utils/util.py:261-286 semantics on a name->shape table (conv W~N(0,.02), conv b=0,
    BN gamma~N(1,.02), BN beta~N(0,1), running stats 0/1, alpha=1)."""
    sd: SD = {}
    for k, shp in shapes.items():
        if k.endswith('running_mean'):
            sd[k] = torch.zeros(shp)
        elif k.endswith('running_var'):
            sd[k] = torch.ones(shp)
        elif k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('alpha'):
            sd[k] = torch.ones(shp)
        elif len(shp) == 4:
            sd[k] = torch.randn(shp, generator=gen) * 0.02
        elif k.endswith('.weight'):
            sd[k] = 1.0 + torch.randn(shp, generator=gen) * 0.02
        elif k.endswith('.bias'):
            # conv bias -> 0 ; BN bias -> N(0,1).  A conv bias belongs to a module whose weight is 4-d.
            wk = k[:-5] + '.weight'
            if wk in shapes and len(shapes[wk]) == 4:
                sd[k] = torch.zeros(shp)
            else:
                sd[k] = torch.randn(shp, generator=gen)
    return sd


def unet_shapes(ngf: int, num_downs: int = 8, in_nc: int = 3, out_nc: int = 3) -> Dict[str, Tuple[int, ...]]:
    """Parameter/buffer shapes of UnetGenertor(filter_cfgs=None).  models/Pix2Pix.py:85-127"""
    D = num_downs
    width = [min(ngf * 2 ** d, ngf * 8) for d in range(D)]       # e[d] channels
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def bn(key, c):
        shp[key + '.weight'] = (c,)
        shp[key + '.bias'] = (c,)
        shp[key + '.running_mean'] = (c,)
        shp[key + '.running_var'] = (c,)
        shp[key + '.num_batches_tracked'] = ()

    shp['model.model.0.weight'] = (width[0], in_nc, 4, 4)
    for d in range(1, D):
        p = unet_block_prefix(d)
        shp[p + '.model.1.weight'] = (width[d], width[d - 1], 4, 4)
        if d < D - 1:
            bn(p + '.model.2', width[d])
    # up path: depth d up-conv maps (cat of depth d+1 | e[D-1]) -> width[d-1]
    p = unet_block_prefix(D - 1)
    shp[p + '.model.3.weight'] = (width[D - 1], width[D - 2], 4, 4)
    bn(p + '.model.4', width[D - 2])
    for d in range(D - 2, 0, -1):
        p = unet_block_prefix(d)
        shp[p + '.model.5.weight'] = (2 * width[d], width[d - 1], 4, 4)
        bn(p + '.model.6', width[d - 1])
    shp['model.model.3.weight'] = (2 * width[0], out_nc, 4, 4)
    shp['model.model.3.bias'] = (out_nc,)
    return shp


def unet_shapes_cfg(filter_cfgs: Sequence[int], channel_cfgs: Sequence[int], in_nc: int = 3, out_nc: int = 3):
    """Shapes of UnetGenertor(filter_cfgs, channel_cfgs), num_downs = 8 (models/Pix2Pix.py:85-127; index layout in SURVEY.md
    Appendix A.1).  A block whose widths are zero is not built (:87, :97); the blocks that are nest by position, and when the
    innermost kind is gone the last one is a loop block around Identity (:59-67)."""
    f, c = [int(v) for v in filter_cfgs], [int(v) for v in channel_cfgs]
    D = 8
    present = [0, 1, 2, 3] + [d for d in (4, 5, 6) if f[d] != 0 and f[15 - d] != 0] + ([7] if f[7] != 0 and f[8] != 0 else [])
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def bn(key, ch):
        for sfx, v in (('weight', (ch,)), ('bias', (ch,)), ('running_mean', (ch,)), ('running_var', (ch,)),
                       ('num_batches_tracked', ())):
            shp[key + '.' + sfx] = v
    shp['model.model.0.weight'] = (f[0], in_nc, 4, 4)
    for j in range(1, len(present)):
        d, p = present[j], unet_block_prefix(j)
        shp[p + '.model.1.weight'] = (f[d], c[d - 1], 4, 4)
        if d < D - 1:
            bn(p + '.model.2', f[d])
    for j in range(len(present) - 1, 0, -1):
        d, p = present[j], unet_block_prefix(j)
        if d == D - 1:
            shp[p + '.model.3.weight'] = (c[7], f[8], 4, 4)
            bn(p + '.model.4', f[8])
        else:
            shp[p + '.model.5.weight'] = (c[14 - d], f[15 - d], 4, 4)
            bn(p + '.model.6', f[15 - d])
    shp['model.model.3.weight'] = (c[14], out_nc, 4, 4)
    shp['model.model.3.bias'] = (out_nc,)
    return shp


def patchgan_shapes(ndf: int, in_nc: int = 6, masked: bool = False, n_layers: int = 3, norm: str = 'batch'):
    """norm='instance': the CycleGAN plain D (bias on every conv, no norm parameters)"""
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()
    lay = patchgan_layout(masked, n_layers)
    chans = [ndf * min(2 ** i, 8) for i in range(n_layers + 1)]
    cin = in_nc
    for li, (ci, bi, gi) in enumerate(lay):
        cout = chans[li] if li < len(lay) - 1 else 1
        shp['model.%d.weight' % ci] = (cout, cin, 4, 4)
        if bi is None or norm == 'instance':
            shp['model.%d.bias' % ci] = (cout,)
        else:
            for s, v in (('weight', (cout,)), ('bias', (cout,)), ('running_mean', (cout,)),
                         ('running_var', (cout,)), ('num_batches_tracked', ())):
                shp['model.%d.%s' % (bi, s)] = v
        if gi is not None:
            shp['model.%d.alpha' % gi] = (cout,)
        cin = cout
    return shp


# ----------------------------------------------------------------------------------------------
# the model step
# ----------------------------------------------------------------------------------------------
class Opt:
    """Just the flags the step reads (options/options.py)."""

    def __init__(self, **kw):
        self.ngf = 32
        self.ndf = 128
        self.teacher_ngf = 64
        self.teacher_ndf = 128
        self.num_downs = 8
        self.no_dropout = True
        self.gan_mode = 'hinge'
        self.lambda_L1 = 100.0
        self.lambda_gram = 1e4
        self.lambda_content = 50.0
        self.lambda_weight = 0.0
        self.lambda_scale = 0.0
        self.lr = 2e-4
        self.arch_lr = 1e-4
        self.ema_beta = 1.0
        self.threshold = 0.5
        self.darts_discriminator = True
        self.online_distillation = True
        self.direction = 'AtoB'
        self.lambda_SR_adversarial = 1e-3   # SRGAN
        self.lambda_SR_content = 0.0
        self.lambda_SR_perceptual = 1.0
        self.lambda_A = 10.0            # CycleGAN
        self.lambda_B = 10.0
        self.lambda_identity = 0.5
        self.__dict__.update(kw)


def _is_float_param(k: str) -> bool:
    return k.endswith('.weight') or k.endswith('.bias') or k.endswith('.alpha')


class Pix2PixOracle:
    """One replica of the reference Pix2PixModel reduced to its arithmetic.

    ``G``/``D`` are state-dict-keyed dicts of fp32 tensors (requires_grad toggled per phase);
    ``T`` are the student's 1x1 transform convs (models/Pix2Pix.py:407-409).
    """

    def __init__(self, opt: Opt, G: SD, D: SD, T: Optional[List[Tensor]] = None, masked: bool = False,
                 teacher: Optional['Pix2PixOracle'] = None):
        self.opt = opt
        self.G, self.D, self.T = G, D, (T or [])
        self.masked = masked
        self.teacher = teacher
        self.g_hooks = unet_hook_names(opt.num_downs)      # (resnet backbone: model.9 / 12 / 15 / 18)
        self.d_hooks = ['model.4', 'model.12'] if masked else ['model.3', 'model.9']
        self.g_feats: OrderedDict = OrderedDict()
        self.d_feats: OrderedDict = OrderedDict()
        self.train = True
        self.current_D_arch_diff_loss = 0.0
        self.losses: Dict[str, float] = {}
        self.lr_G = self.lr_D = opt.lr
        self.lr_arch = opt.arch_lr
        self.dropout_masks = None
        # parameter groups, in the reference's optimizer order
        self.G_keys = [k for k in G if _is_float_param(k)]
        self.D_w_keys = [k for k in D if (k.endswith('.weight') or k.endswith('.bias'))]
        self.D_a_keys = [k for k in D if k.endswith('.alpha')]
        self.st_G, self.st_D, self.st_A = {}, {}, {}

    # -- forward pieces ------------------------------------------------------------------
    def netG(self, x):
        if getattr(self.opt, 'backbone', 'unet') == 'resnet':
            return mobile_resnet_forward(self.G, x, features=self.g_feats)
        return unet_forward(self.G, x, self.opt.num_downs, self.train, dropout=not self.opt.no_dropout,
                            dropout_masks=self.dropout_masks, features=self.g_feats)

    def netD(self, x):
        return patchgan_forward(self.D, x, self.masked, self.opt.threshold, self.train,
                                features=self.d_feats, hook_names=self.d_hooks)

    def set_input(self, A: Tensor, B: Tensor):
        """A, B as in the batch dict; direction picks the roles (models/Pix2Pix.py:453-458)."""
        self.in_A, self.in_B = A, B
        self.real_A, self.real_B = (A, B) if self.opt.direction == 'AtoB' else (B, A)

    def forward(self):
        self.fake_B = self.netG(self.real_A)

    def features(self) -> List[Tensor]:
        return list(self.g_feats.values()) + list(self.d_feats.values())

    def _req(self, sd: SD, keys: Sequence[str], flag: bool):
        for k in keys:
            sd[k].requires_grad_(flag)
            if flag:
                sd[k].grad = None

    # -- one iteration (models/Pix2Pix.py:565-583) -----------------------------------------
    def optimize_parameters(self):
        o = self.opt
        if self.teacher is not None:
            self.teacher.set_input(self.in_A, self.in_B)
            self.teacher.optimize_parameters()
            self.targets = [f.detach().clone() for f in self.teacher.features()]

        Gp = self.G_keys
        self._req(self.G, Gp, True)
        for t in self.T:
            t.requires_grad_(True)
            t.grad = None
        self.forward()

        # ---- D step (:464-477) ----
        self._req(self.D, self.D_w_keys, True)
        self._req(self.D, self.D_a_keys, False)
        pred_fake = self.netD(torch.cat((self.real_A, self.fake_B), 1).detach())
        loss_D_fake = gan_loss(o.gan_mode, pred_fake, False, True)
        pred_real = self.netD(torch.cat((self.real_A, self.real_B), 1))
        loss_D_real = gan_loss(o.gan_mode, pred_real, True, True)
        loss_D = (loss_D_fake + loss_D_real) * 0.5
        loss_D.backward()
        adam_step([self.D[k] for k in self.D_w_keys], [self.D[k].grad for k in self.D_w_keys],
                  self.st_D, self.lr_D, (0.5, 0.999))
        self._req(self.D, self.D_w_keys, False)

        # ---- G step (:513-552) ----
        pred_fake = self.netD(torch.cat((self.real_A, self.fake_B), 1))
        loss_G_GAN = gan_loss(o.gan_mode, pred_fake, True, False)
        loss_G_L1 = F.l1_loss(self.fake_B, self.real_B) * o.lambda_L1
        loss_G = loss_G_GAN + loss_G_L1
        self.losses.update(G_GAN=float(loss_G_GAN.detach()), G_L1=float(loss_G_L1.detach()), D_real=float(loss_D_real.detach()),
                           D_fake=float(loss_D_fake.detach()))
        if self.teacher is not None:
            feats = list(self.g_feats.values())
            # teacher D (train mode, weights frozen) on the student's fake: hazard H2
            self.teacher.netD(torch.cat((self.real_A, self.fake_B), 1))
            feats = feats + list(self.teacher.d_feats.values())
            loss_gram = 0.0
            loss_content = 0.0
            for i, f in enumerate(feats):
                if i < 4:
                    f = F.conv2d(f, self.T[i])
                t = self.targets[i]
                loss_gram = loss_gram + rmse(gram(f), gram(t))
                loss_content = loss_content + rmse(f, t)
            loss_gram = o.lambda_gram * loss_gram
            loss_content = o.lambda_content * loss_content
            loss_G = loss_G + loss_gram + loss_content
            self.losses.update(gram=float(loss_gram.detach()), content=float(loss_content.detach()))
            self.dist_feats = [f.detach() for f in feats]
        loss_G.backward()
        # L1 sparsity sub-gradient (:554-563)
        if o.lambda_weight > 0.0:
            for k in Gp:
                if self.G[k].dim() == 4:
                    self.G[k].grad.add_(o.lambda_weight * torch.sign(self.G[k].detach()))
        elif o.lambda_scale > 0.0:
            for k in Gp:
                if k.endswith('.weight') and self.G[k].dim() == 1:
                    self.G[k].grad.add_(o.lambda_scale * torch.sign(self.G[k].detach()))
        params = [self.G[k] for k in Gp] + list(self.T)
        with torch.no_grad():
            adam_step([p for p in params], [p.grad for p in params], self.st_G, self.lr_G, (0.5, 0.999))
        self.fake_B = self.fake_B.detach()
        self._req(self.G, Gp, False)
        for t in self.T:
            t.requires_grad_(False)

    # -- arch step (models/Pix2Pix.py:479-511, 585-593) --------------------------------------
    def _arch_diff(self, is_teacher: bool):
        o = self.opt
        pred_fake = self.netD(torch.cat((self.real_A, self.fake_B), 1).detach())
        self.loss_D_arch_fake = gan_loss(o.gan_mode, pred_fake, False, True)
        fake_real = gan_loss(o.gan_mode, pred_fake, True, False)
        pred_real = self.netD(torch.cat((self.real_A, self.real_B), 1))
        self.loss_D_arch_real = gan_loss(o.gan_mode, pred_real, True, True)
        cur = (fake_real - self.loss_D_arch_fake).abs()
        if is_teacher and float(self.current_D_arch_diff_loss) != 0.0:
            cur = o.ema_beta * cur + (1.0 - o.ema_beta) * self.current_D_arch_diff_loss
        self.current_D_arch_diff_loss = cur
        return cur

    def clipping_mask_alpha(self):
        with torch.no_grad():
            for k in self.D_a_keys:
                self.D[k].clamp_(0, 1)

    def optimizer_netD_arch(self):
        T = self.teacher
        with torch.no_grad():
            self.forward()
            T.set_input(self.in_A, self.in_B)
            T.forward()
        self._req(self.D, self.D_w_keys, False)
        self._req(self.D, self.D_a_keys, True)
        with torch.no_grad():
            t_diff = T._arch_diff(True)
        s_diff = self._arch_diff(False)
        loss = (s_diff - t_diff).abs() + (self.loss_D_arch_real + self.loss_D_arch_fake) * 0.5
        loss.backward()
        with torch.no_grad():
            adam_step([self.D[k] for k in self.D_a_keys], [self.D[k].grad for k in self.D_a_keys],
                      self.st_A, self.lr_arch, (0.9, 0.999))
        self._req(self.D, self.D_a_keys, False)
        self.losses.update(D_arch_diff=float(s_diff.detach()), D_arch=float(loss.detach()), teacher_D_arch_diff=float(t_diff.detach()))


def build_gcc_pair(opt: Opt, seed: int = 0) -> Pix2PixOracle:
    """Student (masked D, transform convs) + online teacher, initialised with the reference's
    init rule from one seeded generator (not the reference's RNG stream: parity tests load the
    reference's own state_dicts from the golden fixtures instead)."""
    g = torch.Generator().manual_seed(seed)
    tG = init_state_dict(unet_shapes(opt.teacher_ngf, opt.num_downs), g)
    tD = init_state_dict(patchgan_shapes(opt.teacher_ndf, 6, False), g)
    sG = init_state_dict(unet_shapes(opt.ngf, opt.num_downs), g)
    sD = init_state_dict(patchgan_shapes(opt.ndf, 6, opt.darts_discriminator), g)
    s_w = [opt.ngf * 2, opt.ngf * 8, opt.ngf * 16, opt.ngf * 4]
    t_w = [opt.teacher_ngf * 2, opt.teacher_ngf * 8, opt.teacher_ngf * 16, opt.teacher_ngf * 4]
    # nn.Conv2d default init (kaiming_uniform a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    T = [(torch.rand((t, s, 1, 1), generator=g) * 2 - 1) / math.sqrt(s) for s, t in zip(s_w, t_w)]
    teacher = Pix2PixOracle(opt, tG, tD, masked=False)
    return Pix2PixOracle(opt, sG, sD, T, masked=opt.darts_discriminator, teacher=teacher)


# ----------------------------------------------------------------------------------------------
# CycleGAN step (models/CycleGAN.py:218-620)
# ----------------------------------------------------------------------------------------------
class ImagePool:
    """utils/image_pool.py:5-54 -- history of generated images; draws from Python's ``random`` exactly as the
    reference does (one uniform per image once the pool is full, one randint when it swaps)."""

    def __init__(self, pool_size: int, rng=None):
        import random as _random
        self.pool_size, self.images, self.rng = pool_size, [], (rng or _random)

    def query(self, images: Tensor) -> Tensor:
        if self.pool_size == 0:
            return images
        out = []
        for img in images:
            img = img.detach().unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(img)
                out.append(img)
            elif self.rng.uniform(0, 1) > 0.5:
                j = self.rng.randint(0, self.pool_size - 1)
                out.append(self.images[j].clone())
                self.images[j] = img
            else:
                out.append(img)
        return torch.cat(out, 0)


class CycleGANOracle:
    """MobileCycleGANModel reduced to its arithmetic.  G / D / T are dicts {'A': ..., 'B': ...}: netG_A maps A->B and
    is judged by netD_A on domain B.  forward() runs each generator once per distinct input (the reference repeats
    G_A(real_A) and G_B(real_B) only to refresh its hooks: same values, and autograd sums the same gradients)."""
    HEAVY = ('model.1', 'model.4', 'model.19', 'model.22')

    def __init__(self, opt: Opt, G: Dict[str, SD], D: Dict[str, SD], T: Optional[Dict[str, List[Tensor]]] = None,
                 masked: bool = False, teacher: Optional['CycleGANOracle'] = None, pool_size: int = 50, rng=None):
        self.opt, self.G, self.D, self.T = opt, G, D, (T or {'A': [], 'B': []})
        self.masked, self.teacher = masked, teacher
        self.d_hooks = ['model.4', 'model.12'] if masked else ['model.3', 'model.9']
        self.g_feats = {'A': OrderedDict(), 'B': OrderedDict()}
        self.d_feats = {'A': OrderedDict(), 'B': OrderedDict()}
        self.pool = {'A': ImagePool(pool_size, rng), 'B': ImagePool(pool_size, rng)}     # keyed by the D that consumes it
        self.cur_diff = {'A': 0.0, 'B': 0.0}
        self.losses: Dict[str, float] = {}
        self.lr_G = self.lr_D = opt.lr
        self.lr_arch = opt.arch_lr
        self.train = True
        self.G_keys = {w: [k for k in G[w] if _is_float_param(k)] for w in 'AB'}
        self.D_w_keys = {w: [k for k in D[w] if k.endswith('.weight') or k.endswith('.bias')] for w in 'AB'}
        self.D_a_keys = {w: [k for k in D[w] if k.endswith('.alpha')] for w in 'AB'}
        self.st_G, self.st_D, self.st_A = {}, {}, {}

    def netG(self, w, x, hook=False):
        return mobile_resnet_forward(self.G[w], x, features=self.g_feats[w] if hook else None)

    def netD(self, w, x):
        return patchgan_forward(self.D[w], x, self.masked, self.opt.threshold, self.train, features=self.d_feats[w],
                                hook_names=self.d_hooks)

    def set_input(self, A: Tensor, B: Tensor):
        self.in_A, self.in_B = A, B
        self.real_A, self.real_B = (A, B) if self.opt.direction == 'AtoB' else (B, A)

    def forward(self):                                                       # :366-380
        self.fake_B = self.netG('A', self.real_A, hook=True)
        self.rec_A = self.netG('B', self.fake_B)
        self.fake_A = self.netG('B', self.real_B, hook=True)
        self.rec_B = self.netG('A', self.fake_A)
        self.idt_A = self.netG('A', self.real_B)
        self.idt_B = self.netG('B', self.real_A)

    def features(self, w) -> List[Tensor]:
        return list(self.g_feats[w].values()) + list(self.d_feats[w].values())

    def _req(self, sd, keys, flag):
        for k in keys:
            sd[k].requires_grad_(flag)
            if flag:
                sd[k].grad = None

    def optimize_parameters(self):                                           # :571-590
        o = self.opt
        T = self.teacher
        if T is not None:
            T.set_input(self.in_A, self.in_B)
            T.optimize_parameters()
            self.targets = {w: [f.detach().clone() for f in T.features(w)] for w in 'AB'}
        for w in 'AB':
            self._req(self.G[w], self.G_keys[w], True)
            for t in self.T[w]:
                t.requires_grad_(True)
                t.grad = None
            self._req(self.D[w], self.D_w_keys[w] + self.D_a_keys[w], False)
        self.forward()
        # ---- generators (:480-546); criterionGAN(pred, True) keeps for_discriminator's default True
        L = {}
        L['idt_A'] = F.l1_loss(self.idt_A, self.real_B) * o.lambda_B * o.lambda_identity
        L['idt_B'] = F.l1_loss(self.idt_B, self.real_A) * o.lambda_A * o.lambda_identity
        L['G_A'] = gan_loss(o.gan_mode, self.netD('A', self.fake_B), True, True)
        L['G_B'] = gan_loss(o.gan_mode, self.netD('B', self.fake_A), True, True)
        L['cycle_A'] = F.l1_loss(self.rec_A, self.real_A) * o.lambda_A
        L['cycle_B'] = F.l1_loss(self.rec_B, self.real_B) * o.lambda_B
        loss_G = L['G_A'] + L['G_B'] + L['cycle_A'] + L['cycle_B'] + L['idt_A'] + L['idt_B']
        if T is not None:
            # the teacher's discriminators see the student's fakes detached (:497-498): their two features enter the
            # loss value but carry no gradient
            T.netD('A', self.fake_B.detach())
            T.netD('B', self.fake_A.detach())
            for w, fake, tfake in (('A', self.fake_B, T.fake_B), ('B', self.fake_A, T.fake_A)):
                feats = list(self.g_feats[w].values()) + [f.detach() for f in T.d_feats[w].values()]
                gram_l = content_l = l1_l = 0.0
                for i, f in enumerate(feats):
                    if i < 4:
                        f = F.conv2d(f, self.T[w][i])
                    t = self.targets[w][i]
                    gram_l = gram_l + F.mse_loss(gram(f), gram(t))
                    content_l = content_l + F.mse_loss(f, t)
                    l1_l = l1_l + F.l1_loss(fake, tfake.detach())           # inside the loop: counted once per feature
                L['gram_' + w], L['content_' + w], L['L1_' + w] = o.lambda_gram * gram_l, o.lambda_content * content_l, o.lambda_L1 * l1_l
                loss_G = loss_G + L['gram_' + w] + L['content_' + w] + L['L1_' + w]
        loss_G.backward()
        if o.lambda_weight > 0.0:                                            # L1_sparsity (:548-569)
            for w in 'AB':
                for k in self.G_keys[w]:
                    if self.G[w][k].dim() == 4:
                        name = k[:-len('.weight')]
                        mult = 1.0 if name not in self.HEAVY else (1000.0 if name == 'model.19' else 2.0)
                        self.G[w][k].grad.add_(o.lambda_weight * mult * torch.sign(self.G[w][k].detach()))
        params = [self.G[w][k] for w in 'AB' for k in self.G_keys[w]] + [t for w in 'AB' for t in self.T[w]]
        adam_step(params, [p.grad for p in params], self.st_G, self.lr_G, (0.5, 0.999))
        for w in 'AB':
            self._req(self.G[w], self.G_keys[w], False)
            for t in self.T[w]:
                t.requires_grad_(False)
        for n in ('fake_A', 'fake_B', 'rec_A', 'rec_B', 'idt_A', 'idt_B'):
            setattr(self, n, getattr(self, n).detach())
        # ---- discriminators (:382-405): real first, then the pooled fake
        for w, real, fake in (('A', self.real_B, self.fake_B), ('B', self.real_A, self.fake_A)):
            self._req(self.D[w], self.D_w_keys[w], True)
            pooled = self.pool[w].query(fake)
            l_real = gan_loss(o.gan_mode, self.netD(w, real), True, True)
            l_fake = gan_loss(o.gan_mode, self.netD(w, pooled.detach()), False, True)
            L['D_' + w] = (l_real + l_fake) * 0.5
            L['D_' + w].backward()
        params = [self.D[w][k] for w in 'AB' for k in self.D_w_keys[w]]
        adam_step(params, [p.grad for p in params], self.st_D, self.lr_D, (0.5, 0.999))
        for w in 'AB':
            self._req(self.D[w], self.D_w_keys[w], False)
        self.losses.update({k: float(v.detach()) for k, v in L.items()})

    def get_D_arch_diff(self, is_teacher: bool):                             # :417-459
        o = self.opt
        self.arch = {}
        new = {}
        for w, fake, real in (('A', self.fake_B, self.real_B), ('B', self.fake_A, self.real_A)):
            pf = self.netD(w, fake.detach())
            l_fake = gan_loss(o.gan_mode, pf, False, True)
            l_fake_real = gan_loss(o.gan_mode, pf, True, False)
            l_real = gan_loss(o.gan_mode, self.netD(w, real), True, True)
            self.arch[w] = (l_fake, l_real)
            new[w] = (l_fake_real - l_fake).abs()
        if is_teacher and float(self.cur_diff['A']) != 0.0:       # one test (on A) switches the EMA for both
            new = {w: o.ema_beta * new[w] + (1.0 - o.ema_beta) * self.cur_diff[w] for w in 'AB'}
        self.cur_diff = new
        return new['A'], new['B']

    def clipping_mask_alpha(self):
        with torch.no_grad():
            for w in 'AB':
                for k in self.D_a_keys[w]:
                    self.D[w][k].clamp_(0, 1)

    def optimizer_netD_arch(self):                                           # :592-600, 407-415
        T = self.teacher
        with torch.no_grad():
            self.forward()
            T.set_input(self.in_A, self.in_B)
            T.forward()
            t_diff = T.get_D_arch_diff(True)
        for w in 'AB':
            self._req(self.D[w], self.D_w_keys[w], False)
            self._req(self.D[w], self.D_a_keys[w], True)
        s_diff = self.get_D_arch_diff(False)
        for i, w in enumerate('AB'):
            loss = (s_diff[i] - t_diff[i]).abs() + (self.arch[w][0] + self.arch[w][1]) * 0.5
            loss.backward()
            self.losses.update({'D_arch_diff_' + w: float(s_diff[i].detach()), 'D_arch_' + w: float(loss.detach()),
                                'teacher_netD_%s_arch_diff' % w: float(t_diff[i])})
        params = [self.D[w][k] for w in 'AB' for k in self.D_a_keys[w]]
        adam_step(params, [p.grad for p in params], self.st_A, self.lr_arch, (0.9, 0.999))
        for w in 'AB':
            self._req(self.D[w], self.D_a_keys[w], False)


# ----------------------------------------------------------------------------------------------
# SAGAN (models/SAGAN.py): spectral norm, self attention, generator / discriminators, the step
# ----------------------------------------------------------------------------------------------
def spectral_weight(sd: SD, prefix: str) -> Tensor:
    """SpectralNorm._update_u_v (models/SAGAN.py:25-38): one power iteration that overwrites u, v in ``sd`` (in eval
    mode too), then W = W_bar / sigma with sigma = u^T W_bar v differentiable in W_bar only."""
    w = sd[prefix + '.weight_bar']
    u, v = sd[prefix + '.weight_u'], sd[prefix + '.weight_v']
    h = w.shape[0]
    wm = w.reshape(h, -1)
    # ``.data =`` exactly as the reference: u, v keep their identity, so a graph built by an EARLIER call of the same
    # layer (D on the real batch, then D on the fake batch, one backward) evaluates d sigma / d W_bar = u v^T with the
    # vectors of the LATEST power iteration, while its forward value used the sigma of its own call.  The HIP path
    # reproduces this: the gradient transform reads the live u, v and the context's own sigma.
    t = torch.mv(wm.detach().t(), u)
    v.data = t / (t.norm() + 1e-12)
    t = torch.mv(wm.detach(), v)
    u.data = t / (t.norm() + 1e-12)
    sigma = u.dot(wm.mv(v))
    return w / sigma


def self_attention(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """Self_Attn.forward (models/SAGAN.py:72-104): energy = q^T k over the H*W positions, softmax over keys,
    out = v . attention^T, y = gamma * out + x"""
    b, c, hh, ww = x.shape
    n = hh * ww
    q = _q(F.conv2d(x, _qw(sd[prefix + '.query_conv.weight']), sd[prefix + '.query_conv.bias'])).reshape(b, -1, n)
    k = _q(F.conv2d(x, _qw(sd[prefix + '.key_conv.weight']), sd[prefix + '.key_conv.bias'])).reshape(b, -1, n)
    v = _q(F.conv2d(x, _qw(sd[prefix + '.value_conv.weight']), sd[prefix + '.value_conv.bias'])).reshape(b, -1, n)
    attn = torch.softmax(torch.bmm(q.permute(0, 2, 1), k), dim=-1)
    # EMULATE_BF16: the HIP kernels feed the probabilities to the matrix cores in bf16 (the map itself stays fp32)
    out = _q(torch.bmm(v, _q(attn).permute(0, 2, 1)).reshape(b, c, hh, ww))
    return _q(sd[prefix + '.gamma'] * out + x)


def sagan_generator_forward(sd: SD, z: Tensor, train: bool = True, features: Optional[OrderedDict] = None) -> Tensor:
    """Generator.forward (models/SAGAN.py:159-170), image_size 64: l1..l4 = SN(ConvTranspose) + BN + ReLU, attention
    after l3 and l4, last = ConvTranspose + Tanh.  Hooks: 'l2' (post-ReLU), 'attn2'."""
    h = _q(z.reshape(z.shape[0], z.shape[1], 1, 1))
    for i, (stride, pad) in enumerate(((1, 0), (2, 1), (2, 1), (2, 1)), start=1):
        w = _qw(spectral_weight(sd, 'l%d.0.module' % i))
        h = _q(F.conv_transpose2d(h, w, sd['l%d.0.module.bias' % i], stride=stride, padding=pad))
        h = _q(F.relu(batch_norm(sd, 'l%d.1' % i, h, train)))
        if i == 2 and features is not None:
            features['l2'] = h
        if i == 3:
            h = self_attention(sd, 'attn1', h)
    h = self_attention(sd, 'attn2', h)
    if features is not None:
        features['attn2'] = h
    return _q(torch.tanh(F.conv_transpose2d(h, _qw(sd['last.0.weight']), sd['last.0.bias'], stride=2, padding=1)))


def sagan_discriminator_forward(sd: SD, x: Tensor, masked: bool = False, threshold: float = 0.5,
                                features: Optional[OrderedDict] = None) -> Tensor:
    """Discriminator / MaskDiscriminator.forward (models/SAGAN.py:172-274): l1..l4 = SN(Conv k4 s2 p1) [+ gate] +
    LeakyReLU(0.1), attention after l3 and l4, last = Conv k4 (4x4 -> 1x1), squeezed.  Hooks: 'l2', 'attn2'."""
    h = _q(x)
    for i in range(1, 5):
        w = _qw(spectral_weight(sd, 'l%d.0.module' % i))
        h = _q(F.conv2d(h, w, sd['l%d.0.module.bias' % i], stride=2, padding=1))
        if masked:
            h = gate(h, sd['l%d.1.alpha' % i], threshold)
        h = _q(F.leaky_relu(h, 0.1))
        if i == 2 and features is not None:
            features['l2'] = h
        if i == 3:
            h = self_attention(sd, 'attn1', h)
    h = self_attention(sd, 'attn2', h)
    if features is not None:
        features['attn2'] = h
    return F.conv2d(h, _qw(sd['last.0.weight']), sd['last.0.bias']).squeeze()


def _attn_shapes(shp, name, c):
    for n, co in (('query_conv', c // 8), ('key_conv', c // 8), ('value_conv', c)):
        shp['%s.%s.weight' % (name, n)] = (co, c, 1, 1)
        shp['%s.%s.bias' % (name, n)] = (co,)


def sagan_generator_shapes(ngf: int = 64, z_dim: int = 128, filter_cfgs: Optional[Sequence[int]] = None):
    """state_dict shapes in the reference's order: per SN layer bias, weight_u, weight_v, weight_bar (the order
    _make_params registers them after deleting 'weight'), then the BN; l4 is registered before l1 (models/SAGAN.py:141-152)"""
    w = list(filter_cfgs) if filter_cfgs is not None else [ngf * 8, ngf * 4, ngf * 2, ngf]
    cin = [z_dim] + w[:3]
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def layer(i):
        p = 'l%d.0.module' % i
        ws = (cin[i - 1], w[i - 1], 4, 4)
        shp[p + '.bias'] = (w[i - 1],)
        shp[p + '.weight_u'] = (ws[0],)
        shp[p + '.weight_v'] = (ws[1] * 16,)
        shp[p + '.weight_bar'] = ws
        for sfx, v in (('weight', (w[i - 1],)), ('bias', (w[i - 1],)), ('running_mean', (w[i - 1],)),
                       ('running_var', (w[i - 1],)), ('num_batches_tracked', ())):
            shp['l%d.1.%s' % (i, sfx)] = v
    for i in (4, 1, 2, 3):
        layer(i)
    shp['last.0.weight'] = (w[3], 3, 4, 4)
    shp['last.0.bias'] = (3,)
    for name, c in (('attn1', w[2]), ('attn2', w[3])):
        shp[name + '.gamma'] = (1,)
        _attn_shapes(shp, name, c)
    return shp


def sagan_discriminator_shapes(ndf: int = 64, masked: bool = False):
    w = [ndf, ndf * 2, ndf * 4, ndf * 8]
    cin = [3] + w[:3]
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def layer(i):
        p = 'l%d.0.module' % i
        shp[p + '.bias'] = (w[i - 1],)
        shp[p + '.weight_u'] = (w[i - 1],)
        shp[p + '.weight_v'] = (cin[i - 1] * 16,)
        shp[p + '.weight_bar'] = (w[i - 1], cin[i - 1], 4, 4)
        if masked:
            shp['l%d.1.alpha' % i] = (w[i - 1],)
    for i in (4, 1, 2, 3):
        layer(i)
    shp['last.0.weight'] = (1, w[3], 4, 4)
    shp['last.0.bias'] = (1,)
    for name, c in (('attn1', w[2]), ('attn2', w[3])):
        shp[name + '.gamma'] = (1,)
        _attn_shapes(shp, name, c)
    return shp


class SAGANOracle:
    """SAGANModel reduced to its arithmetic (models/SAGAN.py:276-560).  Adam betas (0, 0.9), D learning rate x4.
    ``dup``: parameter names the reference lists twice in an optimizer (hazard H5: SpectralNorm / Self_Attn containers
    and their children are both collected when distillation / the masked D build the lists) -- torch's Adam then
    applies two sequential updates per step to those tensors, with the same gradient."""

    def __init__(self, opt: Opt, G: SD, D: SD, T: Optional[List[Tensor]] = None, masked: bool = False,
                 teacher: Optional['SAGANOracle'] = None):
        self.opt, self.G, self.D, self.T = opt, G, D, (T or [])
        self.masked, self.teacher = masked, teacher
        self.g_feats, self.d_feats = OrderedDict(), OrderedDict()
        self.train = True
        self.cur_diff = 0.0
        self.losses: Dict[str, float] = {}
        self.lr_G, self.lr_D, self.lr_arch = opt.lr, opt.lr * 4, opt.arch_lr
        trainable = lambda k: (k.endswith('.weight') or k.endswith('.bias') or k.endswith('.weight_bar') or k.endswith('.gamma'))
        self.G_keys = [k for k in G if trainable(k)]
        # set_requires_grad(netD, True) (models/SAGAN.py:513) also switches on the discriminator's power-iteration
        # vectors u, v: they receive gradients through sigma = u . (W_bar v) and Adam updates in every D step (the
        # generator's u, v never do)
        self.D_w_keys = [k for k in D if not k.endswith('.alpha')]
        self.D_a_keys = [k for k in D if k.endswith('.alpha')]
        dup = lambda k: ('.module.' in k) or ('_conv.' in k)
        self.G_dup = [k for k in self.G_keys if dup(k)] if self.T else []
        self.D_dup = [k for k in self.D_w_keys if dup(k)] if masked else []
        self.st_G, self.st_D, self.st_A = {}, {}, {}

    def netG(self, z):
        return sagan_generator_forward(self.G, z, self.train, features=self.g_feats)

    def netD(self, x):
        return sagan_discriminator_forward(self.D, x, self.masked, 0.5, features=self.d_feats)   # H6: --threshold ignored

    def set_input(self, z, real):
        self.z, self.real_img = z, real

    def forward(self):
        self.fake_img = self.netG(self.z)

    def features(self):
        return list(self.g_feats.values()) + list(self.d_feats.values())

    def _req(self, sd, keys, flag):
        for k in keys:
            sd[k].requires_grad_(flag)
            if flag:
                sd[k].grad = None

    @staticmethod
    def _adam_dup(sd, keys, dup, extra, state, lr):
        """one Adam step over keys (+ extra tensors); tensors listed twice get a second sequential update"""
        params = [sd[k] for k in keys] + list(extra)
        names = list(keys) + [None] * len(extra)
        if 'per' not in state:
            state['per'] = [dict() for _ in params]
        for p, n, st in zip(params, names, state['per']):
            if p.grad is None:
                continue
            for _ in range(2 if n in dup else 1):
                adam_step([p], [p.grad], st, lr, (0.0, 0.9))

    def optimize_parameters(self):
        o, T = self.opt, self.teacher
        if T is not None:
            T.set_input(self.z, self.real_img)
            T.optimize_parameters()
            self.targets = [f.detach().clone() for f in T.features()]
        self._req(self.G, self.G_keys, True)
        for t in self.T:
            t.requires_grad_(True)
            t.grad = None
        self.forward()
        # ---- D (:370-381): real first, then the detached fake; no 0.5
        self._req(self.D, self.D_w_keys, True)
        self._req(self.D, self.D_a_keys, False)
        l_real = gan_loss(o.gan_mode, self.netD(self.real_img), True, True)
        l_fake = gan_loss(o.gan_mode, self.netD(self.fake_img.detach()), False, True)
        (l_fake + l_real).backward()
        self._adam_dup(self.D, self.D_w_keys, self.D_dup, [], self.st_D, self.lr_D)
        self._req(self.D, self.D_w_keys, False)
        # ---- G (:460-494)
        l_gan = gan_loss(o.gan_mode, self.netD(self.fake_img), True, False)
        loss_G = l_gan
        self.losses.update(G_GAN=float(l_gan.detach()), D_real=float(l_real.detach()), D_fake=float(l_fake.detach()))
        if T is not None:
            feats = list(self.g_feats.values())
            T.netD(self.fake_img)                     # not detached: gradient reaches the student through the teacher's D
            feats = feats + list(T.d_feats.values())
            l_gram = l_content = 0.0
            for i, f in enumerate(feats):
                if i < 2:
                    f = F.conv2d(f, self.T[i])
                t = self.targets[i]
                l_gram = l_gram + rmse(gram(f), gram(t))
                l_content = l_content + rmse(f, t)
            l_gram, l_content = o.lambda_gram * l_gram, o.lambda_content * l_content
            l_l1 = o.lambda_L1 * F.l1_loss(self.fake_img, T.fake_img.detach())
            loss_G = loss_G + l_gram + l_content + l_l1
            # the reference accumulates the distillation terms IN PLACE into the tensor that is also loss_G_GAN
            # (``self.loss_G = self.loss_G_GAN`` then ``+=``, :465-488): the reported G_GAN is the whole generator loss
            self.losses.update(gram=float(l_gram.detach()), content=float(l_content.detach()), L1=float(l_l1.detach()),
                               G_GAN=float(loss_G.detach()))
        loss_G.backward()
        if o.lambda_weight > 0.0:                     # L1_sparsity (:496-506): convs that still own a '.weight'
            for k in self.G_keys:
                if k.endswith('.weight') and self.G[k].dim() == 4:
                    self.G[k].grad.add_(o.lambda_weight * torch.sign(self.G[k].detach()))
        elif o.lambda_scale > 0.0:
            for k in self.G_keys:
                if k.endswith('.weight') and self.G[k].dim() == 1:
                    self.G[k].grad.add_(o.lambda_scale * torch.sign(self.G[k].detach()))
        self._adam_dup(self.G, self.G_keys, self.G_dup, self.T, self.st_G, self.lr_G)
        self.fake_img = self.fake_img.detach()
        self._req(self.G, self.G_keys, False)
        for t in self.T:
            t.requires_grad_(False)

    def get_D_arch_diff(self, is_teacher):
        o = self.opt
        pf = self.netD(self.fake_img.detach())
        self.arch_fake = gan_loss(o.gan_mode, pf, False, True)
        fake_real = gan_loss(o.gan_mode, pf, True, False)
        self.arch_real = gan_loss(o.gan_mode, self.netD(self.real_img), True, True)
        cur = (fake_real - self.arch_fake).abs()
        if is_teacher and float(self.cur_diff) != 0.0:
            cur = o.ema_beta * cur + (1.0 - o.ema_beta) * self.cur_diff
        self.cur_diff = cur
        return cur

    def clipping_mask_alpha(self):
        with torch.no_grad():
            for k in self.D_a_keys:
                self.D[k].clamp_(0, 1)

    def optimizer_netD_arch(self):
        T = self.teacher
        with torch.no_grad():
            self.forward()
            T.set_input(self.z, self.real_img)
            T.forward()
            t_diff = T.get_D_arch_diff(True)
        self._req(self.D, self.D_w_keys, False)
        self._req(self.D, self.D_a_keys, True)
        s_diff = self.get_D_arch_diff(False)
        loss = (s_diff - t_diff).abs() + self.arch_real + self.arch_fake          # no 0.5 here (:388-389)
        loss.backward()
        adam_step([self.D[k] for k in self.D_a_keys], [self.D[k].grad for k in self.D_a_keys], self.st_A, self.lr_arch,
                  (0.9, 0.999))
        self._req(self.D, self.D_a_keys, False)
        self.losses.update(D_arch_diff=float(s_diff.detach()), D_arch=float(loss.detach()), teacher_D_arch_diff=float(t_diff))


# ----------------------------------------------------------------------------------------------
# SRGAN (models/SRGAN.py, models/GANLoss.py:95-145, data/sr_dataset.py:15-64)
# ----------------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512)   # up to conv5_4


def to_imagenet_norm(img: Tensor) -> Tensor:
    """convert_image(img, source='[-1, 1]', target='imagenet-norm')"""
    mean = torch.tensor(IMAGENET_MEAN, dtype=img.dtype).reshape(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=img.dtype).reshape(1, 3, 1, 1)
    return ((img + 1.) / 2. - mean) / std


def srresnet_forward(sd: SD, x: Tensor, train: bool = True, features: Optional[OrderedDict] = None,
                     hook_idx: Sequence[int] = (3, 7, 11, 15)) -> Tensor:
    """Generator.forward (models/SRGAN.py:139-199): k9 conv + PReLU | residual blocks (conv3 BN PReLU conv3 BN, +x) |
    conv3 BN + long skip | 2 x (conv3 -> PixelShuffle(2) -> PReLU) | k9 conv + tanh.  Hooks: outputs of residual blocks"""
    def conv(t, name, pad):
        return _q(F.conv2d(t, _qw(sd[name + '.weight']), sd[name + '.bias'], padding=pad))
    h = _q(F.prelu(conv(_q(x), 'conv_block1.conv_block.0', 4), sd['conv_block1.conv_block.1.weight']))
    skip = h
    n_blocks = len({k.split('.')[1] for k in sd if k.startswith('residual_blocks.')})
    for i in range(n_blocks):
        p = 'residual_blocks.%d.' % i
        t = _q(batch_norm(sd, p + 'conv_block1.conv_block.1', conv(h, p + 'conv_block1.conv_block.0', 1), train))
        t = _q(F.prelu(t, sd[p + 'conv_block1.conv_block.2.weight']))
        t = batch_norm(sd, p + 'conv_block2.conv_block.1', conv(t, p + 'conv_block2.conv_block.0', 1), train)
        h = _q(t + h)
        if features is not None and i in hook_idx:
            features['residual_blocks.%d' % i] = h
    t = batch_norm(sd, 'conv_block2.conv_block.1', conv(h, 'conv_block2.conv_block.0', 1), train)
    h = _q(t + skip)
    for j in range(2):
        p = 'subpixel_convolutional_blocks.%d.' % j
        h = _q(F.prelu(F.pixel_shuffle(conv(h, p + 'conv', 1), 2), sd[p + 'prelu.weight']))
    return _q(torch.tanh(F.conv2d(h, _qw(sd['conv_block3.conv_block.0.weight']), sd['conv_block3.conv_block.0.bias'], padding=4)))


def sr_discriminator_forward(sd: SD, x: Tensor, masked: bool = False, threshold: float = 0.5, train: bool = True,
                             features: Optional[OrderedDict] = None) -> Tensor:
    """Discriminator / MaskDiscriminator.forward (models/SRGAN.py:201-297), n_blocks 4: conv3 (stride 1, 2, 1, 2) [+BN]
    [+gate] + LeakyReLU(0.2); global average pool; Linear -> one logit per image.  Hooks: conv_blocks.1 / .3 outputs."""
    h = _q(x)
    n_blocks = len({k.split('.')[1] for k in sd if k.startswith('conv_blocks.')})
    for i in range(n_blocks):
        p = 'conv_blocks.%d.conv_block.' % i
        h = _q(F.conv2d(h, _qw(sd[p + '0.weight']), sd[p + '0.bias'], stride=1 if i % 2 == 0 else 2, padding=1))
        j = 1
        if i != 0:
            h = batch_norm(sd, p + '1', h, train)
            j = 2
        if masked:
            h = gate(h, sd[p + '%d.alpha' % j], threshold)
        h = _q(F.leaky_relu(h, 0.2))
        if features is not None and i in (1, 3):
            features['conv_blocks.%d' % i] = h
    return F.linear(h.mean((2, 3)), sd['fc1.weight'], sd['fc1.bias'])


def vgg_features(sd: SD, x: Tensor, cfg=VGG19_CFG) -> Tensor:
    """TruncatedVGG19(i=5, j=4) (models/GANLoss.py:95-145): vgg19.features[:36] -- conv3 + ReLU ... up to relu5_4; the
    channel widths come from the weights (the golden fixture uses a narrow stand-in: torchvision is not in the image)"""
    h, idx = _q(x), 0
    for c in cfg:
        if c == 'M':
            h = F.max_pool2d(h, 2, 2)
        else:
            h = _q(F.relu(F.conv2d(h, _qw(sd['truncated_vgg19.%d.weight' % idx]), sd['truncated_vgg19.%d.bias' % idx], padding=1)))
            idx += 1
        idx += 1
    return h


def srresnet_shapes(ngf: int = 64, n_blocks: int = 16, filter_cfgs: Optional[Sequence[int]] = None):
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()

    def conv(name, co, ci, k):
        shp[name + '.weight'] = (co, ci, k, k)
        shp[name + '.bias'] = (co,)

    def bn(name, c):
        for sfx, v in (('weight', (c,)), ('bias', (c,)), ('running_mean', (c,)), ('running_var', (c,)), ('num_batches_tracked', ())):
            shp['%s.%s' % (name, sfx)] = v
    conv('conv_block1.conv_block.0', ngf, 3, 9)
    shp['conv_block1.conv_block.1.weight'] = (1,)
    for i in range(n_blocks):
        inner = ngf if filter_cfgs is None else int(filter_cfgs[i])
        p = 'residual_blocks.%d.' % i
        conv(p + 'conv_block1.conv_block.0', inner, ngf, 3)
        bn(p + 'conv_block1.conv_block.1', inner)
        shp[p + 'conv_block1.conv_block.2.weight'] = (1,)
        conv(p + 'conv_block2.conv_block.0', ngf, inner, 3)
        bn(p + 'conv_block2.conv_block.1', ngf)
    conv('conv_block2.conv_block.0', ngf, ngf, 3)
    bn('conv_block2.conv_block.1', ngf)
    for j in range(2):
        conv('subpixel_convolutional_blocks.%d.conv' % j, 4 * ngf, ngf, 3)
        shp['subpixel_convolutional_blocks.%d.prelu.weight' % j] = (1,)
    conv('conv_block3.conv_block.0', 3, ngf, 9)
    return shp


def sr_discriminator_shapes(ndf: int = 64, masked: bool = False, n_blocks: int = 4):
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()
    cin = 3
    for i in range(n_blocks):
        co = (ndf if i == 0 else cin * 2) if i % 2 == 0 else cin
        p = 'conv_blocks.%d.conv_block.' % i
        shp[p + '0.weight'] = (co, cin, 3, 3)
        shp[p + '0.bias'] = (co,)
        j = 1
        if i != 0:
            for sfx, v in (('weight', (co,)), ('bias', (co,)), ('running_mean', (co,)), ('running_var', (co,)), ('num_batches_tracked', ())):
                shp['%s1.%s' % (p, sfx)] = v
            j = 2
        if masked:
            shp['%s%d.alpha' % (p, j)] = (co,)
        cin = co
    shp['fc1.weight'] = (1, cin)
    shp['fc1.bias'] = (1,)
    return shp


def vgg_shapes(widths=VGG19_CFG):
    shp: Dict[str, Tuple[int, ...]] = OrderedDict()
    cin, idx = 3, 0
    for c in widths:
        if c != 'M':
            shp['truncated_vgg19.%d.weight' % idx] = (c, cin, 3, 3)
            shp['truncated_vgg19.%d.bias' % idx] = (c,)
            cin = c
            idx += 1
        idx += 1
    return shp


class SRGANOracle:
    """SRGAN model class reduced to its arithmetic (models/SRGAN.py:299-481).  Generator first, then discriminator;
    vanilla (BCE-with-logits) GAN loss; after backward_G the model's real_hr / fake_hr ARE the ImageNet-normalised
    tensors (:449-450), so the discriminator trains on those.  Under distillation the optimizer is built from
    Conv / BatchNorm / Linear modules only: the PReLU slopes are left out and stay frozen (hazard H5)."""

    def __init__(self, opt: Opt, G: SD, D: SD, V: SD, T: Optional[List[Tensor]] = None, masked: bool = False,
                 teacher: Optional['SRGANOracle'] = None, vgg_cfg=VGG19_CFG):
        self.opt, self.G, self.D, self.V, self.T = opt, G, D, V, (T or [])
        self.masked, self.teacher, self.vgg_cfg = masked, teacher, vgg_cfg
        self.g_feats, self.d_feats = OrderedDict(), OrderedDict()
        self.train = True
        self.cur_diff = 0.0
        self.losses: Dict[str, float] = {}
        self.lr_G = self.lr_D = opt.lr
        self.lr_arch = opt.arch_lr
        is_prelu = lambda k: G[k].shape == (1,) and k.endswith('.weight')
        self.G_keys = [k for k in G if _is_float_param(k) and not (self.T and is_prelu(k))]
        self.D_w_keys = [k for k in D if k.endswith('.weight') or k.endswith('.bias')]
        self.D_a_keys = [k for k in D if k.endswith('.alpha')]
        self.st_G, self.st_D, self.st_A = {}, {}, {}

    def netG(self, x):
        return srresnet_forward(self.G, x, self.train, features=self.g_feats)

    def netD(self, x):
        return sr_discriminator_forward(self.D, x, self.masked, self.opt.threshold, self.train, features=self.d_feats)

    def set_input(self, lr_img, hr_img):
        self.real_lr, self.real_hr = lr_img, hr_img
        self.in_lr, self.in_hr = lr_img, hr_img

    def forward(self):
        self.fake_hr = self.netG(self.real_lr)

    def features(self):
        return list(self.g_feats.values()) + list(self.d_feats.values())

    def _req(self, sd, keys, flag):
        for k in keys:
            sd[k].requires_grad_(flag)
            if flag:
                sd[k].grad = None

    def optimize_parameters(self):
        o, T = self.opt, self.teacher
        if T is not None:
            T.set_input(self.in_lr, self.in_hr)
            T.optimize_parameters()
            self.targets = [f.detach().clone() for f in T.features()]
        self._req(self.G, self.G_keys, True)
        for t in self.T:
            t.requires_grad_(True)
            t.grad = None
        self.forward()
        # ---- generator (:446-480)
        self._req(self.D, self.D_w_keys + self.D_a_keys, False)
        l_content = F.mse_loss(self.fake_hr, self.real_hr) * o.lambda_SR_content
        real_n, fake_n = to_imagenet_norm(self.real_hr), to_imagenet_norm(self.fake_hr)
        l_gan = gan_loss(o.gan_mode, self.netD(fake_n), True, True) * o.lambda_SR_adversarial
        l_perc = F.mse_loss(vgg_features(self.V, fake_n, self.vgg_cfg), vgg_features(self.V, real_n, self.vgg_cfg).detach()) * o.lambda_SR_perceptual
        loss_G = l_content + l_gan + l_perc
        self.losses.update(G_GAN=float(l_gan.detach()), content=float(l_content.detach()), perceptual=float(l_perc.detach()))
        if T is not None:
            feats = list(self.g_feats.values())
            T.netD(fake_n)
            feats = feats + list(T.d_feats.values())
            l_gram = l_c = 0.0
            for i, f in enumerate(feats):
                if i < 4:
                    f = F.conv2d(f, self.T[i])
                t = self.targets[i]
                l_gram = l_gram + rmse(gram(f), gram(t))
                l_c = l_c + rmse(f, t)
            l_gram, l_c = o.lambda_gram * l_gram, o.lambda_content * l_c
            l_l1 = o.lambda_L1 * F.l1_loss(fake_n, T.fake_hr.detach())       # both already ImageNet-normalised
            loss_G = loss_G + l_gram + l_c + l_l1
            self.losses.update(gram=float(l_gram.detach()), content=float(l_c.detach()), L1=float(l_l1.detach()))
        loss_G.backward()
        params = [self.G[k] for k in self.G_keys] + list(self.T)
        adam_step(params, [p.grad for p in params], self.st_G, self.lr_G, (0.9, 0.999))
        self._req(self.G, self.G_keys, False)
        for t in self.T:
            t.requires_grad_(False)
        self.real_hr, self.fake_hr = real_n.detach(), fake_n.detach()
        # ---- discriminator (:378-388): on the normalised images, real first
        self._req(self.D, self.D_w_keys, True)
        l_real = gan_loss(o.gan_mode, self.netD(self.real_hr), True, True)
        l_fake = gan_loss(o.gan_mode, self.netD(self.fake_hr), False, True)
        (l_real + l_fake).backward()
        adam_step([self.D[k] for k in self.D_w_keys], [self.D[k].grad for k in self.D_w_keys], self.st_D, self.lr_D, (0.9, 0.999))
        self._req(self.D, self.D_w_keys, False)
        self.losses.update(D_real=float(l_real.detach()), D_fake=float(l_fake.detach()))

    def get_D_arch_diff(self, is_teacher):
        o = self.opt
        self.real_hr, self.fake_hr = to_imagenet_norm(self.real_hr), to_imagenet_norm(self.fake_hr)
        pf = self.netD(self.fake_hr.detach())
        self.arch_fake = gan_loss(o.gan_mode, pf, False, True)
        fake_real = gan_loss(o.gan_mode, pf, True, False)
        self.arch_real = gan_loss(o.gan_mode, self.netD(self.real_hr), True, True)
        cur = (fake_real - self.arch_fake).abs()
        if is_teacher and float(self.cur_diff) != 0.0:
            cur = o.ema_beta * cur + (1.0 - o.ema_beta) * self.cur_diff
        self.cur_diff = cur
        return cur

    def clipping_mask_alpha(self):
        with torch.no_grad():
            for k in self.D_a_keys:
                self.D[k].clamp_(0, 1)

    def optimizer_netD_arch(self):
        T = self.teacher
        with torch.no_grad():
            self.forward()
            T.set_input(self.in_lr, self.in_hr)
            T.forward()
            t_diff = T.get_D_arch_diff(True)
        self._req(self.D, self.D_w_keys, False)
        self._req(self.D, self.D_a_keys, True)
        s_diff = self.get_D_arch_diff(False)
        loss = (s_diff - t_diff).abs() + self.arch_real + self.arch_fake
        loss.backward()
        adam_step([self.D[k] for k in self.D_a_keys], [self.D[k].grad for k in self.D_a_keys], self.st_A, self.lr_arch,
                  (0.9, 0.999))
        self._req(self.D, self.D_a_keys, False)
        self.losses.update(D_arch_diff=float(s_diff.detach()), D_arch=float(loss.detach()), teacher_D_arch_diff=float(t_diff))


# ----------------------------------------------------------------------------------------------
# prune cfgs (integer contract)
# ----------------------------------------------------------------------------------------------
def _unet_bn_names(num_downs: int = 8) -> List[str]:
    """BatchNorm2d modules of the U-Net in named_modules() order: down norms d=1..D-2 on the way in,
    then up norms from the innermost block outwards."""
    D = num_downs
    names = [unet_block_prefix(d) + '.model.2' for d in range(1, D - 1)]
    names.append(unet_block_prefix(D - 1) + '.model.4')
    names += [unet_block_prefix(d) + '.model.6' for d in range(D - 2, 0, -1)]
    return names


def scale_prune_cfg(G: SD, threshold: float, ngf: int, num_downs: int = 8):
    """models/Pix2Pix.py:823-860  ->  (filter_cfgs, channel_cfgs), both length 15 for D=8."""
    D = num_downs
    f, c = [ngf], [ngf]
    inner_up = unet_block_prefix(D - 1) + '.model.4'
    last_down = unet_block_prefix(D - 2) + '.model.2'
    up_flag, up_num = False, 0
    for name in _unet_bn_names(D):
        cnt = int((G[name + '.weight'] > threshold).sum())
        f.append(cnt)
        if name == inner_up:
            up_flag = True
            if cnt == 0:
                f[-2] = 0
        if up_flag:
            up_num += 1
            if f[-2 * up_num] == 0:
                f[-1] = 0
                cnt = 0
            c.append(cnt + f[-1 - 2 * up_num])
        else:
            c.append(cnt)
        if name == last_down:
            if f[-1] == 0:
                f.append(0)
                c.append(0)
            else:
                f.append(ngf * 8)
                c.append(ngf * 8)
    return f, c


def _unet_conv_names(num_downs: int = 8) -> List[Tuple[str, bool]]:
    """(name, is_transposed) of every Conv2d / ConvTranspose2d in named_modules() order."""
    D = num_downs
    out = [('model.model.0', False)]
    out += [(unet_block_prefix(d) + '.model.1', False) for d in range(1, D)]
    out.append((unet_block_prefix(D - 1) + '.model.3', True))
    out += [(unet_block_prefix(d) + '.model.5', True) for d in range(D - 2, 0, -1)]
    out.append(('model.model.3', True))
    return out


def filter_norms(w: Tensor, transposed: bool) -> Tensor:
    """sum |w| over (1,2,3) for Conv2d, (0,2,3) for ConvTranspose2d.  models/Pix2Pix.py:877-880"""
    return w.abs().sum((0, 2, 3) if transposed else (1, 2, 3))


def norm_prune_cfg(G: SD, threshold: float, ngf: int, num_downs: int = 8):
    """models/Pix2Pix.py:866-898  ->  len(f)=16, len(c)=15 for D=8."""
    f, c = [], []
    up_num = 0
    for name, tr in _unet_conv_names(num_downs):
        cnt = int((filter_norms(G[name + '.weight'], tr) > threshold).sum())
        f.append(cnt)
        if tr:
            up_num += 1
            if name != 'model.model.3':
                c.append(cnt + f[-1 - 2 * up_num])
        else:
            c.append(cnt)
    if f[0] == 0:
        f[0] = ngf
        c[0] = ngf
        c[-1] += ngf
    return f, c


def max_min_bn_scale(G: SD, num_downs: int = 8):
    """models/Pix2Pix.py:754-776 (the 'prunable' list is hard-coded for D=8)."""
    p3 = 'model.model.1.model.3.model.3.model.3.model.3'
    prunable = [p3 + '.model.2', p3 + '.model.3.model.2', p3 + '.model.3.model.3.model.4',
                p3 + '.model.3.model.6', p3 + '.model.6']
    un_max, pr_max, mn = float('inf'), -float('inf'), float('inf')
    for name in _unet_bn_names(num_downs):
        w = G[name + '.weight']
        if name in prunable:
            pr_max = max(float(w.max()), pr_max)
        else:
            un_max = min(float(w.max()), un_max)
        mn = min(float(w.min()), mn)
    return min(pr_max, un_max), mn


def max_min_conv_norm(G: SD, num_downs: int = 8):
    """models/Pix2Pix.py:778-818 (unet branch)."""
    p3 = 'model.model.1.model.3.model.3.model.3.model.3'
    prunable = [p3 + '.model.1', p3 + '.model.3.model.1', p3 + '.model.3.model.3.model.1',
                p3 + '.model.3.model.3.model.3', p3 + '.model.3.model.5', p3 + '.model.5']
    un_max, pr_max, mn = float('inf'), -float('inf'), float('inf')
    for name, tr in _unet_conv_names(num_downs):
        nrm = filter_norms(G[name + '.weight'], tr)
        if name in prunable:
            pr_max = max(float(nrm.max()), pr_max)
        else:
            un_max = min(float(nrm.max()), un_max)
        mn = min(float(nrm.min()), mn)
    return min(pr_max, un_max), mn
