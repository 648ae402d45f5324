"""Pix2Pix GCC model on MI355X -- the reference's ``models/Pix2Pix.py`` surface
(set_input / forward / optimize_parameters / optimizer_netD_arch / ... ) over the HIP engine.

What is kept name-for-name (callers: train.py:84-172, test.py, metric/test_metric.py,
utils/prune_util.py): class names, constructor signatures, ``netG`` / ``netD`` module trees whose
``state_dict`` keys equal the reference's (module paths such as ``model.model.1.model.2.weight``),
``loss_names`` / ``visual_names``, the optimizers list, schedulers, checkpoint dict layout.

What is different: the module trees only own parameters; every tensor operation of the step is a
hand-written gfx950 kernel scheduled by gcc_amd.engine (no autograd graph).  The step is written
out explicitly in the order of models/Pix2Pix.py:565-593 and its helpers (:464-552).
Under data parallelism (torch.distributed initialised by the launcher) gradients are summed over
ranks on RCCL at the five backward boundaries of the iteration.
"""
import copy
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as gdist
from .. import engine, ops
from .. import _lib
from .._lib import GccError
from ..utils import util
from .DifferentiableOp import DifferentiableOP
from ._streams import TeacherStreamMixin


# ------------------------------------------------------------------------------------------------
# parameter-owning module trees (state_dict-compatible with the reference)
# ------------------------------------------------------------------------------------------------
def _attach(root, dotted, leaf):
    parts = dotted.split('.')
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    m.add_module(parts[-1], leaf)


def _bn(c):
    return nn.BatchNorm2d(c, affine=True, track_running_stats=True)


class UnetGenertor(nn.Module):
    """Parameter tree of the reference's UnetGenertor (models/Pix2Pix.py:79-130; spelling kept).
    Block at nesting position j: prefix 'model' (j=0) / 'model.model.1' + '.model.3'*(j-1).

    Pruned cfgs may remove blocks (models/Pix2Pix.py:87, 97): the innermost block (depth 7) when filter_cfgs[7] or [8] is 0,
    a loop block (depths 6, 5, 4) when filter_cfgs[6-i] or [9+i] is 0.  `present` lists the depths that are built, outermost
    first; the nesting (and with it every state_dict key) goes by position in that list, as in the reference where a parent
    simply wraps whatever block was built last.  When the innermost block is absent the last block is a loop block around
    Identity (:59-67): conv, BatchNorm, ReLU, transposed conv, BatchNorm -- `inner_identity`."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=None, use_dropout=False,
                 filter_cfgs=None, channel_cfgs=None):
        super().__init__()
        D = num_downs
        self.num_downs, self.use_dropout = D, use_dropout
        present = list(range(D))
        if filter_cfgs is None:
            wd = [min(ngf * 2 ** d, ngf * 8) for d in range(D)]
            down_in = [input_nc] + wd[:-1]
            up_in = [2 * wd[d] for d in range(D - 1)] + [wd[D - 1]]
            up_out = [output_nc] + wd[:-1]
        else:
            if D != 8:
                raise NotImplementedError('filter_cfgs describe the num_downs=8 generator')
            f, c = [int(v) for v in filter_cfgs], [int(v) for v in channel_cfgs]
            wd = f[:8]
            down_in = [input_nc] + c[:7]
            up_in = [c[14 - d] for d in range(8)]
            up_out = [output_nc] + [f[15 - d] for d in range(1, 8)]
            present = [0, 1, 2, 3] + [d for d in (4, 5, 6) if f[d] != 0 and f[15 - d] != 0] + \
                      ([7] if f[7] != 0 and f[8] != 0 else [])
            if any(wd[d] <= 0 or up_in[d] <= 0 or (d > 0 and up_out[d] <= 0) or (d > 0 and down_in[d] <= 0) for d in present):
                raise ValueError('filter_cfgs / channel_cfgs give a built block a zero width: %r %r' % (f, c))
        self.present = present
        self.inner_identity = present[-1] != D - 1
        # the blocks torch's Dropout sits in: the loop blocks (models/Pix2Pix.py:101-104: depths D-2 .. 4), by position
        self.dropout_positions = [j for j, d in enumerate(present) if 4 <= d <= D - 2] if use_dropout else []
        K = len(present)

        def prefix(j):
            return 'model' if j == 0 else 'model.model.1' + '.model.3' * (j - 1)
        _attach(self, 'model.model.0', nn.Conv2d(down_in[0], wd[0], 4, 2, 1, bias=False))
        for j in range(1, K):
            d, p = present[j], prefix(j)
            _attach(self, p + '.model.1', nn.Conv2d(down_in[d], wd[d], 4, 2, 1, bias=False))
            if d < D - 1:
                _attach(self, p + '.model.2', _bn(wd[d]))
        for j in range(K - 1, 0, -1):
            d, p = present[j], prefix(j)
            if d == D - 1:
                _attach(self, p + '.model.3', nn.ConvTranspose2d(up_in[d], up_out[d], 4, 2, 1, bias=False))
                _attach(self, p + '.model.4', _bn(up_out[d]))
            else:
                _attach(self, p + '.model.5', nn.ConvTranspose2d(up_in[d], up_out[d], 4, 2, 1, bias=False))
                _attach(self, p + '.model.6', _bn(up_out[d]))
        _attach(self, 'model.model.3', nn.ConvTranspose2d(up_in[0], up_out[0], 4, 2, 1, bias=True))
        # what a block hands to the one inside it must be what that block's conv takes, and what comes back up must be what the
        # transposed conv takes (the reference would fail in forward(); here the engine would read garbage)
        for j in range(1, K):
            d, o = present[j], present[j - 1]
            if down_in[d] != wd[o]:
                raise ValueError('block at depth %d takes %d channels but its parent (depth %d) produces %d' % (d, down_in[d], o, wd[o]))
            inner = wd[d] if j == K - 1 else wd[d] + up_out[present[j + 1]]
            if up_in[d] != inner:
                raise ValueError('up conv at depth %d takes %d channels, its input has %d' % (d, up_in[d], inner))
        if up_in[0] != wd[0] + up_out[present[1]]:
            raise ValueError('outermost up conv takes %d channels, its input has %d' % (up_in[0], wd[0] + up_out[present[1]]))

    def forward(self, x):
        raise GccError('UnetGenertor owns parameters only; run it through Pix2PixModel (gcc_amd.engine.UnetEngine)')


class SeparableConv2d(nn.Module):
    """depthwise kxk (+bias) -> InstanceNorm -> pointwise 1x1 (+bias)   (models/Pix2Pix.py:132-145); parameters only"""

    def __init__(self, in_channels, out_channels, kernel_size=3):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, in_channels, kernel_size, groups=in_channels, bias=True),
                                  nn.InstanceNorm2d(in_channels), nn.Conv2d(in_channels, out_channels, 1, bias=True))


class MobileResnetBlock(nn.Module):
    """conv_block slots: 0 pad, 1 separable, 2 IN, 3 ReLU, 4 dropout, 5 pad, 6 separable, 7 IN
    (models/Pix2Pix.py:147-197); out = x + conv_block(x)"""

    def __init__(self, c_in, c_mid, c_out, dropout_rate=0.0):
        super().__init__()
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), SeparableConv2d(c_in, c_mid), nn.InstanceNorm2d(c_mid),
                                        nn.ReLU(True), nn.Dropout(dropout_rate), nn.ReflectionPad2d(1),
                                        SeparableConv2d(c_mid, c_out), nn.InstanceNorm2d(c_out))


class MobileResnetGenerator(nn.Module):
    """Parameter tree of the reference's MobileResnetGenerator (models/Pix2Pix.py:199-265, models/CycleGAN.py:77-138).
    cfg: 23 widths [stem x3, (mid, out) x n_blocks, up x2]; a block whose mid width is 0 is left out, which shifts
    the Sequential indices behind it exactly as in the reference."""

    def __init__(self, input_nc=3, output_nc=3, ngf=64, norm_layer=nn.InstanceNorm2d, dropout_rate=0, n_blocks=9,
                 padding_type='reflect', opt=None, cfg=None):
        super().__init__()
        if norm_layer is not nn.InstanceNorm2d or padding_type != 'reflect' or dropout_rate != 0:
            raise NotImplementedError('the MI355X path implements the configuration the reference trains: InstanceNorm, '
                                      'reflect padding, dropout 0')
        self.opt = opt
        w = [int(v) for v in cfg] if cfg is not None else [ngf, 2 * ngf, 4 * ngf] + [4 * ngf] * (2 * n_blocks) + [2 * ngf, ngf]
        assert len(w) == 5 + 2 * n_blocks, 'cfg must hold %d widths' % (5 + 2 * n_blocks)
        seq = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, w[0], 7, bias=True), nn.InstanceNorm2d(w[0]), nn.ReLU(True)]
        for i in (1, 2):
            seq += [nn.Conv2d(w[i - 1], w[i], 3, stride=2, padding=1, bias=True), nn.InstanceNorm2d(w[i]), nn.ReLU(True)]
        j = 3
        for _ in range(n_blocks):
            if w[j] != 0:
                seq.append(MobileResnetBlock(w[j - 1], w[j], w[j + 1], dropout_rate))
            j += 2
        for i in (j, j + 1):
            seq += [nn.ConvTranspose2d(w[i - 1], w[i], 3, stride=2, padding=1, output_padding=1, bias=True),
                    nn.InstanceNorm2d(w[i]), nn.ReLU(True)]
        seq += [nn.ReflectionPad2d(3), nn.Conv2d(w[j + 1], output_nc, 7, bias=True), nn.Tanh()]
        self.model = nn.Sequential(*seq)

    def forward(self, x):
        raise GccError('MobileResnetGenerator owns parameters only; run it through the model classes '
                       '(gcc_amd.engine.MobileResnetEngine)')


def _patchgan_tree(self, input_nc, ndf, n_layers, masked, threshold, norm='batch'):
    """norm='instance': the CycleGAN plain discriminator (InstanceNorm2d without parameters, a bias on every conv)"""
    inorm = norm == 'instance'
    ch = [ndf * min(2 ** i, 8) for i in range(n_layers + 1)]
    seq = nn.Module()
    self.add_module('model', seq)
    i = 0
    seq.add_module(str(i), nn.Conv2d(input_nc, ch[0], 4, 2, 1))
    i += 2
    if masked:
        seq.add_module(str(i), DifferentiableOP(ch[0], threshold))
        i += 1
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        seq.add_module(str(i), nn.Conv2d(ch[n - 1], ch[n], 4, stride, 1, bias=inorm))
        seq.add_module(str(i + 1), nn.InstanceNorm2d(ch[n]) if inorm else _bn(ch[n]))
        i += 2
        if masked:
            seq.add_module(str(i), DifferentiableOP(ch[n], threshold))
            i += 1
        i += 1        # LeakyReLU slot
    seq.add_module(str(i), nn.Conv2d(ch[n_layers], 1, 4, 1, 1))


class NLayerDiscriminator(nn.Module):
    """PatchGAN parameter tree (models/Pix2Pix.py:267-305): convs at model.0/2/5/8/11, BN at 3/6/9."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3):
        super().__init__()
        _patchgan_tree(self, input_nc, ndf, n_layers, False, 0.5)


class MaskNLayerDiscriminator(nn.Module):
    """Selective-activation PatchGAN (models/Pix2Pix.py:307-348): convs at model.0/3/7/11/15,
    BN at 4/8/12, gates (alpha) at 2/5/9/13."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, threshold=0.5):
        super().__init__()
        _patchgan_tree(self, input_nc, ndf, n_layers, True, threshold)


# ------------------------------------------------------------------------------------------------
class HipAdam(torch.optim.Optimizer):
    """torch.optim.Optimizer facade (so LambdaLR/StepLR schedulers work unchanged) whose step() is
    one multi-tensor gcc_adam_step launch over a FlatParams group."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, l1=None, dup=(), layout=None):
        """dup: parameters (members of params) the reference lists twice in this optimizer (SAGAN, SURVEY.md hazard
        H5): torch's Adam then applies two sequential updates per step to them, with the same gradient and the step
        counter advancing twice -- reproduced by a second plan over those tensors that is stepped twice."""
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        dev = params[0].device
        self.flat = engine.FlatParams(params, dev, layout=layout)
        self.reducer = None         # dist.GradReducer under data parallelism (Pix2PixModel sets it)
        l1 = list(l1) if l1 is not None else [0.0] * len(params)
        dup_ids = {id(p) for p in dup}
        once = [i for i, p in enumerate(params) if id(p) not in dup_ids]
        twice = [i for i, p in enumerate(params) if id(p) in dup_ids]
        pick = lambda idx, seq: [seq[i] for i in idx]
        self.plan = ops.AdamPlan(pick(once, params), pick(once, self.flat.grad_views), dev, l1=pick(once, l1)) if once else None
        self.plan_dup = ops.AdamPlan(pick(twice, params), pick(twice, self.flat.grad_views), dev, l1=pick(twice, l1)) if twice else None

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def set_grad_scale(self, s):
        for plan in (self.plan, self.plan_dup):
            if plan is not None:
                plan.set_grad_scale(s)

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if self.plan is not None:
            self.plan.step(g['lr'], g['betas'], g['eps'])
        if self.plan_dup is not None:
            self.plan_dup.step(g['lr'], g['betas'], g['eps'])
            self.plan_dup.step(g['lr'], g['betas'], g['eps'])


# host-enqueue scheduling: alternate the chunks of two generators (GCC_INTERLEAVE=0: teacher first, then the student)
# GCC_INTERLEAVE: 0 (default) the teacher's whole iteration is enqueued first, then the student's; 1 alternate chunk by chunk;
# n >= 2 alternate the first n chunks only.  Measured (profiles/r02_ab_tables.md): every interleaving is 3.5-6 % SLOWER -- the
# teacher's iteration is the critical path (its end gates the student's backward_G) and both networks' discriminator passes
# fill the chip, so work given to the student early only delays the teacher.
INTERLEAVE = int(os.environ.get('GCC_INTERLEAVE', '0'))
# the online teacher's D(real) pass started early too: 1 = on an auxiliary stream of its own (measured: -7 %, a fifth busy queue on
# four hardware queues); 2 = on the STUDENT's auxiliary stream, in front of the student's pass (no extra queue: round 4 A/B)
TEACHER_EARLY_DREAL = int(os.environ.get('GCC_TEACHER_EARLY_DREAL', '0'))
# the student joins the teacher's stream where the teacher's features and discriminator are final (after the head of the
# teacher's backward_G), not at the end of the teacher's iteration: the teacher's generator backward + Adam + repack
# (small and HBM-bound kernels) then run beside the student's distillation passes instead of in front of them
EARLY_JOIN = os.environ.get('GCC_EARLY_JOIN', '1') != '0'
# GCC_DISTILL_FORK (default 1): the distillation terms on the generator's features run on the auxiliary stream beside the
# teacher discriminator's pass over the student's fake (backward_G's tail): +2.6 % (profiles/r4s_ab_distill_fork.txt)
DISTILL_FORK = os.environ.get('GCC_DISTILL_FORK', '1') != '0'
# GCC_ARCH_FORK (default 1): the architecture step's two discriminator backward passes (fake, real) side by side on two
# streams: +1.9 % (profiles/r4y_ab_arch_fork.txt), same bits (tests/test_replay_gpu.py::test_pix2pix_stream_forks_change_nothing)
ARCH_FORK = os.environ.get('GCC_ARCH_FORK', '1') != '0'
# GCC_ARCH_EARLY=1: the online teacher's part of the architecture step (generator forward + two discriminator forwards over the
# validation batch) starts when the teacher's iteration ends instead of when the student has finished reading the teacher
ARCH_EARLY = os.environ.get('GCC_ARCH_EARLY', '0') == '1'
# GCC_TAIL_HALO_HC (default 1): the teacher discriminator's pass over the student's fake (the student's tail: one or two busy queues)
# takes the 128-column halo tiles for its half-chip launches: +0.3-0.6 % (profiles/r4ao_ab_tail_halo_hc.txt)
TAIL_HALO_HC = os.environ.get('GCC_TAIL_HALO_HC', '1') != '0'
# GCC_ARCH_FREE_EARLY (default 1): in the architecture step the teacher's stream hands its difference scalar over itself and is
# released behind its own part: +0.6 % (profiles/r4ay_ab_arch_free_early.txt), same bits
ARCH_FREE_EARLY = os.environ.get('GCC_ARCH_FREE_EARLY', '1') != '0'
# GCC_DP_TEACHER_UPDATE_EARLY (default 1; data parallelism only): the online teacher's generator update -- wait for its gradient
# buckets, Adam, repack -- is enqueued on the teacher's own stream right behind its backward pass instead of in front of the
# architecture step's teacher forward
DP_TEACHER_UPDATE_EARLY = os.environ.get('GCC_DP_TEACHER_UPDATE_EARLY', '1') != '0'
DP_TEACHER_BUCKETS_LATE = os.environ.get('GCC_DP_TEACHER_BUCKETS_LATE', '1') != '0'


def _step(gen, stream):
    """one chunk of `gen` with `stream` current (None: the current stream); False when the generator is exhausted"""
    try:
        if stream:
            with ops.on_stream(stream):
                next(gen)
        else:
            next(gen)
        return True
    except StopIteration:
        return False


def _drain(gen, stream):
    if stream:
        with ops.on_stream(stream):
            for _ in gen:
                pass
    else:
        for _ in gen:
            pass


def _alternate(tgen, ts, sgen):
    """teacher chunk, student chunk, ... until both are exhausted (tgen may be None)"""
    t_alive, s_alive = tgen is not None, sgen is not None
    n = 0
    while t_alive or s_alive:
        if t_alive:
            t_alive = _step(tgen, ts)
        n += 1
        if INTERLEAVE >= 2 and n >= INTERLEAVE and t_alive:
            _drain(tgen, ts)
            t_alive = False
        if s_alive:
            s_alive = _step(sgen, None)


# ------------------------------------------------------------------------------------------------
class Pix2PixModel(TeacherStreamMixin, nn.Module):

    @property
    def replay_supported(self):
        """gcc_amd.replay: the U-Net's dropout seeds are by-value launch arguments that change every iteration and that nothing
        patches, so an iteration with dropout in it stays on the eager host path (--no_dropout, or the resnet backbone whose
        blocks have Dropout(0), can be recorded: tests/test_replay_gpu.py).  The same holds for the online teacher."""
        def dropout(m):
            return (not m.resnet) and bool(getattr(getattr(m, 'G', None), 'drop_depths', None))
        t = getattr(self, 'teacher_model', None)
        return not dropout(self) and not (t is not None and dropout(t))


    def __init__(self, opt, filter_cfgs=None, channel_cfgs=None):
        super().__init__()
        self.opt = opt
        if len(opt.gpu_ids) == 0 or not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU and gpu_ids >= 0')
        self.device = gdist.local_device(opt)
        ops.lib()                      # fail loudly here if libgcc_hip.so is not built
        self.filter_cfgs, self.channel_cfgs = filter_cfgs, channel_cfgs
        self.loss_names = ['G_GAN', 'G_L1', 'D_real', 'D_fake']
        self.visual_names = ['real_A', 'fake_B', 'real_B']
        self.current_D_arch_diff_loss = 0.0
        self.teacher_model = None

        self.resnet = opt.backbone == 'resnet'
        if self.resnet:
            self.generator_extract_layers = ['model.9', 'model.12', 'model.15', 'model.18']
        else:
            self.generator_extract_layers = ['model.model.1.model.2', 'model.model.1.model.3.model.3.model.2',
                                             'model.model.1.model.3.model.3.model.4', 'model.model.1.model.4']
        self.discriminator_extract_layers = ['model.4', 'model.12'] if opt.darts_discriminator else ['model.3', 'model.9']

        self.optimizers = []
        if self.resnet:
            self.netG = MobileResnetGenerator(input_nc=3, output_nc=3, ngf=opt.ngf, cfg=filter_cfgs, opt=opt)
        else:
            self.netG = UnetGenertor(3, 3, opt.num_downs, ngf=opt.ngf, use_dropout=not opt.no_dropout,
                                     filter_cfgs=filter_cfgs, channel_cfgs=channel_cfgs)
        self.distill = bool(opt.online_distillation or opt.normal_distillation)
        self.transform_convs = []
        if self.distill:
            t_w = [opt.teacher_ngf * 2, opt.teacher_ngf * 8, opt.teacher_ngf * 16, opt.teacher_ngf * 4]
            if self.resnet:         # models/Pix2Pix.py:387-397: the four hooked tensors are all 4*ngf wide
                t_w = [opt.teacher_ngf * 4] * 4
                s_w = [opt.ngf * 4 if filter_cfgs is None else filter_cfgs[2]] * 4
            elif channel_cfgs is None:
                s_w = [opt.ngf * 2, opt.ngf * 8, opt.ngf * 16, opt.ngf * 4]
            else:
                s_w = [channel_cfgs[1], channel_cfgs[3], channel_cfgs[-4], channel_cfgs[-2]]
            self.transform_convs = [nn.Conv2d(s, t, 1, 1, 0, bias=False) for s, t in zip(s_w, t_w)]
        if opt.darts_discriminator:
            self.loss_names += ['D_arch_diff', 'D_arch', 'teacher_D_arch_diff']
            self.netD = MaskNLayerDiscriminator(input_nc=6, ndf=opt.ndf, threshold=opt.threshold)
        else:
            self.netD = NLayerDiscriminator(input_nc=6, ndf=opt.ndf)
        self.init_net()

        # ---- optimizers over flat parameter groups (models/Pix2Pix.py:382,415,430-440)
        dev = self.device
        g_params, g_l1 = [], []
        for m in self.netG.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                for p in m.parameters():
                    g_params.append(p)
                    g_l1.append(opt.lambda_weight if (p.dim() == 4 and opt.lambda_weight > 0.0) else 0.0)
            elif isinstance(m, nn.BatchNorm2d):
                g_params += [m.weight, m.bias]
                g_l1 += [opt.lambda_scale if (opt.lambda_scale > 0.0 and not opt.lambda_weight > 0.0) else 0.0, 0.0]
        for t in self.transform_convs:
            t.to(dev)
            gdist.broadcast_module(t)              # default-initialised from each rank's RNG: replicas must start equal
            g_params.append(t.weight)
            g_l1.append(0.0)
        # storage order of the flat gradient buffers = backward-completion order (transform convs, then the U-Net layer by
        # layer; the PatchGAN from its last conv down): data parallelism all-reduces finished runs while the backward continues
        g_layout = d_layout = None
        if not self.resnet:
            g_layout = ([[t.weight for t in self.transform_convs]] if self.transform_convs else []) + \
                engine.UnetEngine.grad_segments(self.netG, opt.num_downs)
        self.optimizer_G = HipAdam(g_params, lr=opt.lr, betas=(0.5, 0.999), l1=g_l1, layout=g_layout)
        w_params, a_params = [], []
        for m in self.netD.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                w_params += list(m.parameters())
            elif isinstance(m, DifferentiableOP):
                a_params += list(m.parameters())
        self.optimizer_D = HipAdam(w_params, lr=opt.lr, betas=(0.5, 0.999), layout=self._d_layout())
        if opt.darts_discriminator:
            self.optimizer_arch = HipAdam(a_params, lr=opt.arch_lr)
            if opt.arch_lr_step:
                arch_opt = copy.deepcopy(opt)
                arch_opt.lr_policy = 'step'
                arch_opt.lr_decay_iters = opt.n_epochs - 1
                self.arch_scheduler = util.get_scheduler(self.optimizer_arch, arch_opt)

        # ---- engines
        if self.resnet:
            self.G = engine.MobileResnetEngine(self.netG, dev)
            t_split = [0, 0, 0, 0]
        else:
            self.G = engine.UnetEngine(self.netG, opt.num_downs, dev, use_dropout=not opt.no_dropout)
            # the last two hooked features are concat buffers (skip | up path): their channel dimension is split
            t_split = [0, 0, self.G.width[3], self.G.width[1]]
        self.D = engine.PatchGANEngine(self.netD, bool(opt.darts_discriminator), opt.threshold, dev)
        self.D.mask_cache = True          # gate masks are recomputed only after alpha changed (arch step, clip, reload)
        if gdist.buckets_enabled(dev):
            mb = int(os.environ.get('GCC_DP_BUCKET_MB', '32'))
            if g_layout is not None:
                self.optimizer_G.reducer = self.G.reducer = gdist.GradReducer(self.optimizer_G, mb << 20)
                self.G.seg_base = 1 if self.transform_convs else 0
            self.optimizer_D.reducer = self.D.reducer = gdist.GradReducer(self.optimizer_D, mb << 20)
        self.T = [engine.ConvOp(t.weight, None, 1, 1, 0, False, col_split=sp)
                  for t, sp in zip(self.transform_convs, t_split)]
        self.refresh_weights()

        self.optimizers += [self.optimizer_G, self.optimizer_D]
        self.schedulers = [util.get_scheduler(o, opt) for o in self.optimizers]
        if opt.darts_discriminator and opt.arch_lr_step:
            self.schedulers.append(self.arch_scheduler)
        # device scalars: every loss of the iteration lives in one fp32 vector (read on demand)
        self._lossvec = torch.zeros(32, dtype=torch.float32, device=dev)
        self._slot = {n: i for i, n in enumerate(
            ['G_GAN', 'G_L1', 'D_real', 'D_fake', 'D_arch_fake', 'D_arch_fake_real', 'D_arch_real', 'D_arch_diff',
             'D_arch', 'teacher_D_arch_diff', 'arch_c_fr', 'arch_c_f', 'scratch0', 'scratch1', 'scratch2', 'ema_prev'])}
        self._dist_out = torch.zeros((6, 2), dtype=torch.float32, device=dev)
        self._dist_ws = {}
        self._fake_nchw = None
        self._ema_started = False
        self._world = gdist.world_size()
        if self._world > 1:
            engine.OVERLAP_WGRAD = engine.overlap_wgrad_default(self._world)
        self._defer_G_update = False
        self._pending_G = None
        self._comm_group = None        # gdist.chain_group('teacher') once this model runs as an online teacher on its own stream

    # ---------------------------------------------------------------------------------------
    def _l(self, name):
        i = self._slot[name]
        return self._lossvec[i:i + 1]

    def refresh_weights(self):
        """re-derive the bf16 weight packings from the fp32 masters (after init / load / Adam)"""
        self.G.repack()
        self.D.repack()
        self.D.mask_dirty = True
        for t in self.T:
            t.repack()

    def init_net(self):
        self.netG.to(self.device)
        self.netD.to(self.device)
        for m in self.netD.modules():
            if isinstance(m, DifferentiableOP):
                m.threshold = m.threshold.to(self.device)
        util.init_weights(self.netG, init_type='normal', init_gain=0.02)
        util.init_weights(self.netD, init_type='normal', init_gain=0.02)
        gdist.broadcast_module(self.netG)
        gdist.broadcast_module(self.netD)

    # ---------------------------------------------------------------------------------------
    def set_input(self, input):
        self.input = input
        self._note_input(input)
        AtoB = self.opt.direction == 'AtoB'
        self.real_A = input['A' if AtoB else 'B'].to(self.device, torch.float32).contiguous()
        self.real_B = input['B' if AtoB else 'A'].to(self.device, torch.float32).contiguous()
        self.image_paths = [input.get('A_paths' if AtoB else 'B_paths'), input.get('B_paths' if AtoB else 'A_paths')]
        N, _, H, W = self.real_A.shape
        if getattr(self, '_A', None) is None or tuple(self._A.shape) != (N, 3, H, W):
            self._A = ops.new_act(N, 3, H, W, self.device)
            self._B = ops.new_act(N, 3, H, W, self.device)
        ops.nchw_to_nhwc(self.real_A, self._A)
        ops.nchw_to_nhwc(self.real_B, self._B)

    def forward(self, slot=0):
        """fake_B = G(real_A)  (models/Pix2Pix.py:460-462).  slot 1: in a second set of activation buffers (the online teacher's
        architecture-step forward, started while the student still reads the features of the training batch)"""
        self.finish_G_update()
        N, _, H, W = self._A.shape
        if self.resnet:
            c = self.G._ctx(N, H, W, 'main' if slot == 0 else 'slot%d' % slot)
        else:
            c = self.G._ctx(N, H, W, slot)
        ops.nhwc_copy(self._A, 0, c.x_in, 0, 3, cfill=8)
        if self.resnet:
            self._gctx = self.G.forward(c, train=self.netG.training)
        else:
            self._gctx = self.G.forward(N, H, W, train=self.netG.training, slot=slot)
        self._fake = self._gctx.out
        self._fake_nchw = None

    @property
    def fake_B(self):
        if self._fake_nchw is None:
            self._fake_nchw = ops.nhwc_to_nchw(self._fake, 3)
        return self._fake_nchw

    @property
    def Tfake_B(self):
        """the teacher's generated image of the same batch (models/Pix2Pix.py:523; a visual, converted on demand)"""
        return self.teacher_model.fake_B

    # -- helpers ------------------------------------------------------------------------------
    def _pack_pair(self, ctx, second, A=None):
        """ctx.x_in = cat(real_A, second) along channels (3 + 3, zero-filled to 8)"""
        ops.nhwc_pack_pair(self._A if A is None else A, second, ctx.x_in, 3, 3)

    def _d_forward(self, tag, second, refresh=True, A=None, defer_running=False):
        """A: the caller's own copy of real_A (the student's, when it runs the teacher's discriminator over its fake: the
        teacher may already hold the next batch).  defer_running: see PatchGANEngine.forward"""
        N, _, H, W = (self._A if A is None else A).shape
        ctx = self.D.new_ctx(N, H, W, tag)
        self._pack_pair(ctx, second, A)
        self.D.forward(ctx, train=True, refresh=refresh, defer_running=defer_running)
        return ctx

    def _start_real_pass(self, tag):
        """D(real_A, real_B) does not depend on the generator: start it on the auxiliary stream before the generator's
        forward (the student's stream is the iteration's critical path: ~460 mostly small launches, while the other
        streams idle half the time).  The pass keeps its place in the reference's order for everything that is order
        dependent: its BatchNorm running-statistics updates are applied in _take_real_pass(), after D(fake)'s."""
        self._early = getattr(self, '_early', {})
        # the student always; the online teacher only with GCC_TEACHER_EARLY_DREAL=1 (measured: 833 against 893 images/s)
        is_teacher = self.teacher_model is None
        aux = self._aux_stream() if (not is_teacher or (TEACHER_EARLY_DREAL and getattr(self, '_is_online_teacher', False))) else False
        if is_teacher and TEACHER_EARLY_DREAL == 2 and aux:
            aux = getattr(self, '_shared_aux', None) or False
        if not aux:
            return
        main = ops.current_stream()
        self.D.refresh_masks() if self.D.masked else None
        ops.wait_stream(aux, main)
        with ops.on_stream(aux):
            N, _, H, W = self._A.shape
            ctx = self.D.new_ctx(N, H, W, tag)
            self._pack_pair(ctx, self._B)
            self.D.forward(ctx, train=True, defer_running=True, refresh=False)
        self._early[tag] = (ctx, aux)

    def _take_real_pass(self, tag):
        """the context of D(real) -- from the auxiliary stream if it was started early, else computed here"""
        got = getattr(self, '_early', {}).pop(tag, None)
        if got is None:
            return self._d_forward(tag, self._B)
        ctx, aux = got
        ops.wait_stream(ops.current_stream(), aux)
        self.D.apply_deferred_running(ctx)
        return ctx

    def _d_layout(self):
        """PatchGAN weight parameters in backward-completion order (engine.PatchGANEngine.grad_segments, by module index)"""
        mods = [m for m in self.netD.model.children() if isinstance(m, (nn.Conv2d, nn.BatchNorm2d))]
        segs, cur = [], []
        for m in mods:                       # forward order: conv, [bn], conv, [bn] ... -> one segment per conv (+ its bn)
            if isinstance(m, nn.Conv2d) and cur:
                segs.append(cur)
                cur = []
            cur = (list(m.parameters()) + cur) if isinstance(m, nn.BatchNorm2d) else cur + list(m.parameters())
        segs.append(cur)
        return segs[::-1]

    def _allreduce(self, optimizer):
        """gradient exchange before an optimizer step: buckets the backward pass already launched are waited for, the rest
        is reduced here"""
        if getattr(optimizer, 'reducer', None) is not None:
            optimizer.reducer.finish()
        else:
            gdist.all_reduce_grads(optimizer, group=self._comm_group)

    # -- D step (models/Pix2Pix.py:464-477) --------------------------------------------------------
    def backward_D(self):
        for _ in self._backward_D_steps():
            pass

    def _backward_D_steps(self):
        """backward_D in host-enqueue chunks (one discriminator pass each): the iteration's scheduler alternates the
        chunks of the student with those of the online teacher, so that both streams have work queued at the same time"""
        mode = self.opt.gan_mode
        early = 'd_real' in getattr(self, '_early', {})
        cf = self._d_forward('d_fake', self._fake, refresh=not early)
        yield
        cr = self._take_real_pass('d_real')
        gp = self.D.grad_pred_buffer(cf)
        if self.optimizer_D.reducer is not None:
            self.optimizer_D.reducer.begin()
        ops.gan_loss(mode, cf.pred, False, True, self._l('D_fake'), dpred=gp, grad_weight=0.5)
        yield
        self.D.backward(cf, wgrad=True, need_dx=False)
        yield
        ops.gan_loss(mode, cr.pred, True, True, self._l('D_real'), dpred=gp, grad_weight=0.5)
        self.D.reduce_now = True             # the second pass completes the gradients: its finished layers are exchanged
        self.D.backward(cr, wgrad=True, need_dx=False)
        self.D.reduce_now = False

    # -- G step (models/Pix2Pix.py:513-552) --------------------------------------------------------
    def backward_G(self, ts=None):
        """ts: the online teacher's stream -- joined only where its features are first read (the student's own GAN / L1
        terms and the discriminator's data gradient do not depend on it)"""
        for _ in self._backward_G_head_steps():
            pass
        self._backward_G_tail(ts)

    def _backward_G_head_steps(self):
        """the part of backward_G that does not read the teacher: D(fake) with the updated discriminator, the GAN / L1
        terms and the discriminator's data gradient"""
        opt, mode = self.opt, self.opt.gan_mode
        gc = self._gctx
        cg = self._d_forward('g_fake', self._fake)
        self._dctx_g = cg
        ops.gan_loss(mode, cg.pred, True, False, self._l('G_GAN'), dpred=self.D.grad_pred_buffer(cg))
        yield
        dx = self.D.backward(cg, wgrad=False, need_dx=True)
        ops.l1_loss(self._fake, self._B, self._l('G_L1'), weight=opt.lambda_L1, da=gc.g_out)
        ops.nhwc_add(dx, 3, gc.g_out, 0, 3)

    def _backward_G_tail(self, ts=None):
        opt = self.opt
        gc = self._gctx
        g_feat = None
        if self.optimizer_G.reducer is not None:
            self.optimizer_G.reducer.begin()
        if ts:
            ev = getattr(self.teacher_model, '_head_done', None) if EARLY_JOIN else None
            if ev is not None:
                ops.wait_event(ops.current_stream(), ev)
            else:
                ops.wait_stream(ops.current_stream(), ts)
        if self.distill:
            aux = self._aux_stream() if DISTILL_FORK else False
            if aux:
                # the generator's own terms on the auxiliary stream, beside the teacher discriminator's forward / backward over
                # the student's fake: this stretch of the step has one busy queue otherwise (GCC_DISTILL_FORK=0: in line).
                # Forking the whole block earlier -- beside the student's discriminator step -- loses 4.7 %: the step is
                # chip-bound there (profiles/r4s_ab_distill_fork.txt)
                ops.wait_stream(aux, ops.current_stream())
                with ops.on_stream(aux):
                    g_feat = self._distill_generator_terms(gc)
            dx2 = self._distill_teacher_d_terms()
            if not aux:
                g_feat = self._distill_generator_terms(gc)
            if aux:
                ops.wait_stream(ops.current_stream(), aux)
            ops.nhwc_add(dx2, 3, gc.g_out, 0, 3)
            self._mark_teacher_free()
            if self.optimizer_G.reducer is not None:
                self.optimizer_G.reducer.segment_done(0)          # the transform convs' gradients (joined above)
        self.G.backward(gc, g_feat=g_feat, wgrad=True)

    def _distill_generator_terms(self, gc):
        """the four distillation terms on the generator's own features (transform conv, gram / content terms, their gradients
        back through the transform; models/Pix2Pix.py:520-533): independent of the teacher discriminator's pass over the
        student's fake.  Returns the gradients w.r.t. the hooked features; runs on the current stream."""
        opt = self.opt
        gfe = self.G.features(gc)
        N = gfe[0].shape[0]
        tf, dtf = [], []
        for i in range(4):
            f = gfe[i]
            buf = self._tbuf(i, N, self.T[i].rows, f.shape[2], f.shape[3])
            self.T[i].forward(f, buf[0])
            tf.append(buf[0])
            dtf.append(buf[1])
        for i in range(4):
            f, t = tf[i], self.target_distillation_features[i]
            ws = self._dws(i, N, f.shape[1], f.shape[2] * f.shape[3])
            ops.distill_fwd(f, t, self._dist_out[i], ws)
            ops.distill_bwd(f, t, opt.lambda_gram, opt.lambda_content, dtf[i], ws)
        out = []
        for i in range(4):
            self.T[i].backward_weight(gfe[i], dtf[i])
            gbuf = self._tbuf(10 + i, N, gfe[i].shape[1], gfe[i].shape[2], gfe[i].shape[3])[0]
            self.T[i].backward_data(dtf[i], gbuf)
            out.append(gbuf)
        ops.SideStream.get(self.device).join()
        return out

    def _distill_teacher_d_terms(self):
        """teacher D (train mode, frozen) on the student's fake, the two terms on its features (:531-533) and their gradient
        back to the image; returns that gradient's buffer (channels 3.. of the pair).  Runs on the current stream."""
        opt = self.opt
        T = self.teacher_model
        # this pass runs where the step has one or two busy queues (the student's tail): its half-chip launches (L3 forward, L4
        # data gradient: 128 workgroups of 256 x 256) take the 128-column halo tiles of the 'alone' plan (GCC_TAIL_HALO_HC)
        if not TAIL_HALO_HC:
            return self._distill_teacher_d_terms_body(T, opt)
        with ops.plan_override(halo_hc=1):
            return self._distill_teacher_d_terms_body(T, opt)

    def _distill_teacher_d_terms_body(self, T, opt):
        ct = T._d_forward('on_student', self._fake, A=self._A)
        dfe = T.D.features(ct)
        N = dfe[0].shape[0]
        dtd = [self._tbuf(4 + j, N, dfe[j].shape[1], dfe[j].shape[2], dfe[j].shape[3])[1] for j in range(2)]
        for j in range(2):
            f, t = dfe[j], self.target_distillation_features[4 + j]
            ws = self._dws(4 + j, N, f.shape[1], f.shape[2] * f.shape[3])
            ops.distill_fwd(f, t, self._dist_out[4 + j], ws)
            ops.distill_bwd(f, t, opt.lambda_gram, opt.lambda_content, dtd[j], ws)
        return T.D.backward(ct, has_pred_grad=False, g_feat=dtd, wgrad=False, need_dx=True)

    def _tbuf(self, i, N, C, H, W):
        key = ('t', i, N, C, H, W)
        if key not in self._dist_ws:
            self._dist_ws[key] = (ops.new_act(N, C, H, W, self.device), ops.new_act(N, C, H, W, self.device))
        return self._dist_ws[key]

    def _dws(self, i, N, C, HW):
        key = ('w', i, N, C, HW)
        need = ops.distill_workspace_bytes(N, C, HW)       # depends on the weight-gradient split plan (tuning options)
        buf = self._dist_ws.get(key)
        if buf is None or buf.numel() < need:
            buf = self._dist_ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    # -- one iteration (models/Pix2Pix.py:565-583) ----------------------------------------------------
    def optimize_parameters(self):
        """models/Pix2Pix.py:565-583.  The online teacher's whole iteration (its own stream) is enqueued first, then the
        student's forward + discriminator step + the teacher-independent head of backward_G; both are generators of
        host-enqueue chunks so that the order can be chosen (GCC_INTERLEAVE, see above: teacher first is the fastest)."""
        self.finish_G_update()
        ts, tgen = None, None
        if self.opt.online_distillation:
            T = self.teacher_model
            T._defer_G_update = True
            T._is_online_teacher = True
            T._shared_aux = self._aux_stream() if TEACHER_EARLY_DREAL == 2 else None
            ts = self._teacher_stream()
            T._own_stream = bool(ts)
            if ts and self._world > 1 and T._comm_group is None and not getattr(T, '_comm_group_set', False):
                # the teacher's chain runs beside this one: its gradient buckets travel on a communicator of their own
                T._comm_group_set = True
                T._comm_group = gdist.chain_group('teacher')
                for o in (T.optimizer_G, T.optimizer_D):
                    if getattr(o, 'reducer', None) is not None:
                        o.reducer.set_group(T._comm_group)
            if ts:
                self._release_teacher_stream(ts)                 # after the last launch that reads the teacher's buffers
            tgen = T._iteration_steps(self.input)
            if not ts or not INTERLEAVE:
                _drain(tgen, ts)
                tgen = None
        sgen = self._pre_join_steps()
        _alternate(tgen, ts, sgen)
        if ts and getattr(self.teacher_model, '_held_G', False):
            # the student's discriminator buckets are issued: now the teacher generator's, and its update, on the teacher's stream
            self.teacher_model._held_G = False
            with ops.on_stream(ts):
                self.teacher_model.finish_G_update()
        if self.opt.online_distillation:
            # the reference clones; here the teacher's activation buffers of this iteration are
            # simply not overwritten before the student consumes them (separate contexts)
            self.target_distillation_features = self.teacher_model.get_distillation_features()
        self._backward_G_tail(ts)
        if self._defer_G_update and self._world > 1:
            # online teacher under data parallelism: its generator is not read again before the arch
            # step, so its (largest, 218 MB) gradient bucket is reduced while the student's whole
            # iteration runs; finish_G_update() applies it.  Same arithmetic, later in stream order.
            self._pending_G = self.optimizer_G.reducer or gdist.all_reduce_grads(self.optimizer_G, async_op=True, group=self._comm_group)
            return
        self._allreduce(self.optimizer_G)
        self._apply_G_update()

    def _pre_join_steps(self):
        """the student's iteration up to the point where the teacher's features are first read"""
        self._start_real_pass('d_real')
        yield
        self.forward()
        yield
        self.optimizer_D.zero_grad()
        yield from self._backward_D_steps()
        self._allreduce(self.optimizer_D)
        self.optimizer_D.step()
        self.D.repack()
        self.optimizer_G.zero_grad()
        yield
        yield from self._backward_G_head_steps()

    def _iteration_steps(self, input):
        """set_input + optimize_parameters of a model without a teacher of its own (the online teacher), as host-enqueue
        chunks for the student's scheduler"""
        self.set_input(input)
        self.finish_G_update()
        yield
        yield from self._pre_join_steps()
        # everything a student reads from this model is final here: the generator's features (forward), the discriminator's
        # weights (its Adam step + repack) and features (the D(fake) pass of backward_G's head)
        if getattr(self, '_head_done', None) is None:
            self._head_done = ops.Event()
        self._head_done.record()
        yield
        # GCC_DP_TEACHER_BUCKETS_LATE (default 1; data parallelism, teacher on its own stream): the teacher generator's gradient
        # buckets are NOT issued during its backward pass.  A communicator runs its collectives in issue order and the host
        # issues this whole iteration before the student's: buckets issued here would sit in front of the student's discriminator
        # buckets and make the student's main stream wait for the END of this backward pass at its D step's finish() (1.6 ms,
        # profiles/r5_dp_one_rank.txt).  The student issues them -- and the update behind them -- on this stream once its own
        # discriminator buckets are out (optimize_parameters); nothing reads this generator's weights before the arch step.
        red = self.optimizer_G.reducer
        hold = bool(DP_TEACHER_BUCKETS_LATE and self._defer_G_update and self._world > 1 and getattr(self, '_own_stream', False))
        if hold and red is not None:
            red.enabled = False
        try:
            self._backward_G_tail(None)
        finally:
            if red is not None:
                red.enabled = True
        if hold:
            self._pending_G = red or gdist.Deferred(self.optimizer_G, self._comm_group)
            self._held_G = True
            return
        if self._defer_G_update and self._world > 1:
            self._pending_G = self.optimizer_G.reducer or gdist.all_reduce_grads(self.optimizer_G, async_op=True, group=self._comm_group)
            if DP_TEACHER_UPDATE_EARLY and getattr(self, '_own_stream', False):
                # On a stream of its own (the online teacher's) nothing else is queued behind this iteration until the
                # architecture step: the wait for the buckets, Adam and the repack go here, where the stream idles, and not in
                # front of the arch step's forward, where the student's arch backward waits for them (round 5: +0.5 ms on the
                # critical chain of a data-parallel rank, profiles/r5_dp_one_rank.txt).  The main stream never waits for this.
                self.finish_G_update()
            return
        self._allreduce(self.optimizer_G)
        self._apply_G_update()

    def _apply_G_update(self):
        self.optimizer_G.step()          # L1_sparsity() (:554-563) is fused into the Adam kernel
        if self.T and isinstance(self.G, engine.UnetEngine):     # generator + transform convs: one repack launch (on the iteration's critical chain)
            if getattr(self, '_g_pack', None) is None:
                self._g_pack = ops.PackPlan(list(self.G.convs()) + list(self.T), self.device)
            self._g_pack.run()
        else:
            self.G.repack()
            for t in self.T:
                t.repack()

    def finish_G_update(self):
        if self._pending_G is not None:
            self._pending_G.wait()
            self._pending_G = None
            self._apply_G_update()

    # -- architecture step (models/Pix2Pix.py:479-511, 585-593) -----------------------------------------
    def get_D_arch_diff(self, isTeacher=False, defer_running=False):
        """three hinge terms on one fake / real pair; the |.| difference (EMA'd for the teacher).
        defer_running: both passes leave the BatchNorm running statistics alone (the caller replays the updates where the
        passes belong in the reference's order: PatchGANEngine.apply_deferred_running)"""
        mode = self.opt.gan_mode
        early = 'a_real' in getattr(self, '_early', {})
        if defer_running:
            cf = self._d_forward('a_fake', self._fake, defer_running=True)
            cr = self._d_forward('a_real', self._B, refresh=False, defer_running=True)
        else:
            cf = self._d_forward('a_fake', self._fake, refresh=not early)
            cr = self._take_real_pass('a_real')
        ops.gan_loss(mode, cf.pred, False, True, self._l('D_arch_fake'))
        ops.gan_loss(mode, cf.pred, True, False, self._l('D_arch_fake_real'))
        ops.gan_loss(mode, cr.pred, True, True, self._l('D_arch_real'))
        out = self._l('teacher_D_arch_diff' if isTeacher else 'D_arch_diff')
        w = 1.0
        if isTeacher and self._world > 1:
            # the teacher's difference feeds an EMA that every replica must hold identically -> the two hinge
            # means are summed over ranks (in place: the teacher logs neither) and the 1/world goes into the scalar op
            i = self._slot['D_arch_fake']
            assert self._slot['D_arch_fake_real'] == i + 1
            gdist.all_reduce_sum(self._lossvec[i:i + 2], group=self._comm_group)
            w = 1.0 / gdist.world_size()
        if isTeacher and self._ema_started:
            b = float(self.opt.ema_beta)
            ops.scalar_op(1, self._l('D_arch_fake_real'), self._l('D_arch_fake'), out, c=out, k0=b * w, k1=1.0 - b)
        elif w != 1.0:
            ops.scalar_op(1, self._l('D_arch_fake_real'), self._l('D_arch_fake'), out, c=out, k0=w, k1=0.0)
        else:
            ops.scalar_op(0, self._l('D_arch_fake_real'), self._l('D_arch_fake'), out)
        self._ema_started = True
        self.current_D_arch_diff_loss = out
        return cf, cr

    def backward_D_arch(self, ts=None):
        T = self.teacher_model
        if not ts:
            T.get_D_arch_diff(isTeacher=True)
        cf, cr = self.get_D_arch_diff(isTeacher=False)
        if ts:
            ops.wait_stream(ops.current_stream(), ts)          # the teacher's difference was computed on its stream
        if not (ts and ARCH_FREE_EARLY):
            ops.scalar_op(2, T._l('teacher_D_arch_diff'), T._l('teacher_D_arch_diff'), self._l('teacher_D_arch_diff'), k0=0.0)
            self._mark_teacher_free()
        # (else: the teacher's own stream copied the scalar and released itself at the end of its part, optimizer_netD_arch)
        # loss_D_arch = |d_S - d_T| + (L_real + L_fake)/2 ; coefficients of the three hinge gradients
        ops.arch_coeffs(self._l('D_arch_fake_real'), self._l('D_arch_fake'), self._l('D_arch_real'),
                        self._l('teacher_D_arch_diff'), self._l('D_arch'), self._l('arch_c_fr'), self._l('arch_c_f'))
        mode = self.opt.gan_mode
        gp = self.D.grad_pred_buffer(cf)
        ops.gan_loss(mode, cf.pred, True, False, self._l('scratch0'), dpred=gp, weight_dev=self._l('arch_c_fr'))
        ops.gan_loss(mode, cf.pred, False, True, self._l('scratch1'), dpred=gp, weight_dev=self._l('arch_c_f'),
                     dpred_accumulate=True)
        flat = getattr(self.optimizer_arch, 'flat', None)
        aux = self._aux_stream() if (ARCH_FORK and flat is not None and isinstance(self.D, engine.PatchGANEngine)) else False
        if aux:
            # the two passes are independent up to the sum of their alpha gradients: the pass over the real pair runs on the
            # auxiliary stream in gradient buffers of its own and accumulates into a zeroed copy of the arch optimizer's flat
            # gradient, which is added afterwards -- (0 + fake) + (0 + real), the bits of fake-then-real (each layer's sum
            # arrives in one add: bnact_bwd_finalize)
            side = self._arch_side_grads(flat)
            ops.wait_stream(aux, ops.current_stream())
            with ops.on_stream(aux):
                ops.fill(side[0], 0.0)
                gp1 = self.D.grad_pred_buffer(cr, slot=1)
                ops.gan_loss(mode, cr.pred, True, True, self._l('scratch2'), dpred=gp1, grad_weight=0.5)
                self.D.backward(cr, wgrad=False, agrad=True, need_dx=False, gslot=1, dalpha=side[1])
            self.D.backward(cf, wgrad=False, agrad=True, need_dx=False)
            ops.wait_stream(ops.current_stream(), aux)
            ops.add_f32_(flat.grads, side[0])
            return
        self.D.backward(cf, wgrad=False, agrad=True, need_dx=False)
        ops.gan_loss(mode, cr.pred, True, True, self._l('scratch2'), dpred=gp, grad_weight=0.5)
        self.D.backward(cr, wgrad=False, agrad=True, need_dx=False)

    def _arch_side_grads(self, flat):
        """(buffer shaped like the arch optimizer's flat gradient, {layer: view of it where that layer's alpha.grad sits})"""
        got = getattr(self, '_arch_side', None)
        if got is None or got[0].numel() != flat.grads.numel():
            buf = torch.zeros_like(flat.grads)
            views = {}
            for li, gate in enumerate(self.D.gate):
                if gate is not None:
                    g = gate.alpha.grad
                    off = (g.data_ptr() - flat.grads.data_ptr()) // 4
                    assert 0 <= off and off + g.numel() <= buf.numel() and g.is_contiguous()
                    views[li] = buf[off:off + g.numel()].view_as(g)
            got = self._arch_side = (buf, views)
        return got

    def optimizer_netD_arch(self):
        T = self.teacher_model
        ts = self._teacher_stream()

        free = getattr(self, '_teacher_free', None)
        early = bool(ts) and ARCH_EARLY and free is not None and isinstance(T.D, engine.PatchGANEngine)

        def hand_over():
            # GCC_ARCH_FREE_EARLY: the last thing the student reads of the teacher in this step is its difference scalar -- the
            # teacher's stream copies it into the student's loss vector itself and is free from here (its next iteration then
            # starts behind its own architecture-step part, not behind the student's two discriminator forwards)
            if ts and ARCH_FREE_EARLY:
                ops.scalar_op(2, T._l('teacher_D_arch_diff'), T._l('teacher_D_arch_diff'), self._l('teacher_D_arch_diff'), k0=0.0)
                self._mark_teacher_free()

        def teacher_part():
            T.finish_G_update()
            T.set_input(self.input)
            yield
            if not early:
                T.forward()
                yield
                T.get_D_arch_diff(isTeacher=True)
                hand_over()
                return
            # GCC_ARCH_EARLY: started when the teacher's own iteration ends, not when the student has finished reading the
            # teacher (its stream idles ~2 ms there).  What the student still reads stays untouched: the generator's features
            # (this forward runs in a second set of activation buffers), the discriminator contexts of the training batch
            # (other tags), real_A (the student packs its own copy); the BatchNorm running statistics of the teacher's
            # discriminator are updated where these two passes belong -- after the student's pass over it
            T.forward(slot=1)
            yield
            cf, cr = T.get_D_arch_diff(isTeacher=True, defer_running=True)
            ops.wait_event(ops.current_stream(), free)
            T.D.apply_deferred_running(cf)
            T.D.apply_deferred_running(cr)
            hand_over()

        def student_part():
            if ts:
                self._start_real_pass('a_real')
                yield
            self.forward()

        if ts:
            self._release_teacher_stream(ts, wait_free=not early)
            tgen = teacher_part()
            if not INTERLEAVE:
                _drain(tgen, ts)
                tgen = None
            _alternate(tgen, ts, student_part())
        else:
            _drain(student_part(), None)
            T.finish_G_update()
            T.set_input(self.input)
            T.forward()
        self.optimizer_arch.zero_grad()
        self.backward_D_arch(ts)
        self._allreduce(self.optimizer_arch)
        self.optimizer_arch.step()
        self.D.mask_dirty = True

    def clipping_mask_alpha(self):
        flat = getattr(getattr(self, 'optimizer_arch', None), 'flat', None)
        if flat is not None:
            ops.clamp_(flat.values, 0.0, 1.0)          # every alpha lives in the arch optimizer's flat buffer: one launch
        else:
            for m in self.netD.modules():
                if isinstance(m, DifferentiableOP):
                    m.clip_alpha()
        self.D.mask_dirty = True

    # -- bookkeeping surface ----------------------------------------------------------------------
    def print_sparse_info(self, logger):
        for name, m in self.named_modules():
            if isinstance(m, DifferentiableOP):
                mask = m.get_current_mask()
                logger.info('%s sparsity ratio: %.2f' % (name, float((mask == 0.0).sum()) / mask.numel()))

    def adaptive_ema_beta(self, epoch):
        self.opt.ema_beta = 1.0 - epoch / (self.opt.n_epochs + self.opt.n_epochs_decay)

    def update_learning_rate(self, epoch):
        for s in self.schedulers:
            s.step()
        self.adaptive_ema_beta(epoch)
        lr = self.optimizers[0].param_groups[0]['lr']
        print('learning rate = %.7f\tema beta = %.7f' % (lr, self.opt.ema_beta))

    def set_requires_grad(self, nets, requires_grad=False):
        for net in (nets if isinstance(nets, list) else [nets]):
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad

    def save_models(self, epoch, save_dir, fid=None, isbest=False, direction='AtoB'):
        self.finish_G_update()
        if gdist.rank() != 0:
            return
        util.mkdirs(save_dir)
        ckpt = {'G': _portable(self.netG.state_dict()), 'D': _portable(self.netD.state_dict()), 'epoch': epoch,
                'cfg': (self.filter_cfgs, self.channel_cfgs), 'fid': fid}
        name = 'model_best_%s.pth' % direction if isbest else 'model_%d.pth' % epoch
        torch.save(ckpt, os.path.join(save_dir, name))

    def load_models(self, load_path, load_discriminator=True):
        ckpt = torch.load(load_path, map_location='cpu')
        self.netG.load_state_dict(ckpt['G'])
        if load_discriminator:
            self.netD.load_state_dict(ckpt['D'])
        self.refresh_weights()
        print('loading the model from %s' % load_path)
        return ckpt['fid'], float('inf')

    def model_train(self):
        self.netG.train()
        self.netD.train()

    def model_eval(self):
        self.netG.eval()
        self.netD.eval()

    def get_current_visuals(self):
        ret = OrderedDict()
        for name in self.visual_names:
            ret[name] = getattr(self, name)
        return ret

    def get_current_losses(self):
        """host read of the device loss scalars (the only sync of the iteration; print_freq cadence)"""
        v = self._lossvec.cpu()
        d = self._dist_out.cpu()
        ret = OrderedDict()
        for name in self.loss_names:
            if name == 'content':
                val = self.opt.lambda_content * float(d[:, 1].sum())
            elif name == 'gram':
                val = self.opt.lambda_gram * float(d[:, 0].sum())
            else:
                val = float(v[self._slot[name]])
            ret[name] = val
        if self._world > 1:
            ret = gdist.mean_dict(ret, self.device)
        return ret

    def init_distillation(self):
        if self.distill:
            if self.opt.lambda_content > 0.0:
                self.loss_names.append('content')
            if self.opt.lambda_gram > 0.0:
                self.loss_names.append('gram')
            self.visual_names.append('Tfake_B')

    def get_distillation_features(self):
        """4 generator features (hooked modules of :366-369) + 2 discriminator features, as the
        tensors the reference's hooks end up holding (post in-place activation, hazard H1)"""
        return self.G.features(self._gctx) + self.D.features(self._dctx_g)

    def get_cfg(self):
        return self.filter_cfgs, self.channel_cfgs

    # -- pruning cfgs (integer logic on host copies of the weights; models/Pix2Pix.py:742-952) -----------
    def _bn_sd(self):
        return {k: v.detach().cpu() for k, v in self.netG.state_dict().items() if k.endswith('.weight') and v.dim() == 1}

    def scale_prune_cfg(self, threshold):
        from ..utils import prune_util
        if torch.is_tensor(threshold):
            threshold = threshold.detach().cpu()
        return prune_util.scale_prune_cfg(self._bn_sd(), threshold, self.opt.ngf, self.opt.num_downs)

    def scale_prune(self, threshold):
        f, c = self.scale_prune_cfg(threshold)
        return Pix2PixModel(self.opt, filter_cfgs=f, channel_cfgs=c)

    def norm_prune(self, threshold):
        from ..utils import prune_util
        f, c = prune_util.norm_prune_cfg(self.netG, threshold, self.opt.ngf)
        return Pix2PixModel(self.opt, filter_cfgs=f, channel_cfgs=c)

    def resnet_prune(self, threshold):
        from ..utils import prune_util
        return Pix2PixModel(self.opt, filter_cfgs=prune_util.resnet_prune_cfg(self.netG, threshold, 'union'))

    def prune(self, threshold, lottery_path=None):
        """lottery_path is accepted and ignored: the reference's prune_util passes it (utils/prune_util.py:57) to a
        method that does not take it (models/Pix2Pix.py:742; SURVEY.md hazard H7)"""
        if self.opt.backbone == 'resnet':
            return self.resnet_prune(threshold)
        if self.opt.scale_prune:
            return self.scale_prune(threshold)
        if self.opt.norm_prune:
            return self.norm_prune(threshold)
        raise NotImplementedError('only scale and norm pruning are supported!!!')

    def max_min_bn_scale(self):
        from ..utils import prune_util
        return prune_util.max_min_bn_scale(self._bn_sd(), self.opt.num_downs)

    def max_min_conv_norm(self):
        from ..utils import prune_util
        if self.opt.backbone == 'resnet':
            return prune_util.max_min_conv_norm_resnet(self.netG, 'union')
        return prune_util.max_min_conv_norm_unet(self.netG)


def _portable(sd):
    """NCHW-contiguous fp32 CPU copies, as a reference checkpoint stores them"""
    return OrderedDict((k, v.detach().to('cpu').contiguous()) for k, v in sd.items())
