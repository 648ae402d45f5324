"""SRGAN GCC model on MI355X -- the reference's ``models/SRGAN.py`` surface over the HIP engine.

Kept name-for-name: ConvolutionalBlock / SubPixelConvolutionalBlock / ResidualBlock / Generator / Discriminator /
MaskDiscriminator / SRGAN and TruncatedVGG19 (models/GANLoss.py:95-145), their constructor signatures and state_dict
keys, ``loss_names`` / ``visual_names``, the checkpoint dict layout ('G', 'D', 'epoch', 'cfg', 'psnr').

Reference behaviour reproduced on purpose:
  * generator first, then discriminator (models/SRGAN.py:483-503); vanilla (BCE-with-logits) GAN loss;
  * backward_G replaces ``real_hr`` / ``fake_hr`` by their ImageNet-normalised versions (:449-450): the discriminator
    step, the teacher's discriminator and the L1-to-teacher term all see normalised images;
  * under distillation the optimizer is built from Conv / BatchNorm / Linear modules only, so the PReLU slopes are
    frozen (SURVEY.md hazard H5); 'content' in the logged losses is then the distillation term;
  * the truncated VGG19 needs torchvision's ImageNet weights, which the reference downloads.  Here they are read from
    ``$GCC_VGG19_WEIGHTS`` (a torchvision ``vgg19`` state_dict: keys ``features.N.weight/bias``); without it the model
    refuses to build unless ``GCC_VGG19_RANDOM=1`` asks for seeded random weights (benchmarks / tests with a stand-in).
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as gdist
from .. import engine, ops
from .._lib import GccError, check
from ..utils import util
from .DifferentiableOp import DifferentiableOP
from .Pix2Pix import HipAdam, _portable
from ._streams import TeacherStreamMixin

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')


# GCC_SR_FORK (default 1): backward_G's VGG chain on the auxiliary stream beside the discriminator's pass over the fake
SR_FORK = os.environ.get('GCC_SR_FORK', '1') != '0'
TEACHER_CHAIN_WGRAD = os.environ.get('GCC_SR_TEACHER_CHAIN_WGRAD', '0') == '1'

class ConvolutionalBlock(nn.Module):
    """conv (padding k // 2) [+ BatchNorm] [+ DifferentiableOP] [+ PReLU | LeakyReLU(0.2) | Tanh]  (models/SRGAN.py:15-70)"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, batch_norm=False, activation=None, mask=False,
                 threshold=0.5):
        super().__init__()
        if activation is not None:
            activation = activation.lower()
            assert activation in {'prelu', 'leakyrelu', 'tanh'}
        layers = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2)]
        if batch_norm is True:
            layers.append(nn.BatchNorm2d(out_channels))
        if mask:
            layers.append(DifferentiableOP(out_channels, threshold=threshold))
        if activation == 'prelu':
            layers.append(nn.PReLU())
        elif activation == 'leakyrelu':
            layers.append(nn.LeakyReLU(0.2))
        elif activation == 'tanh':
            layers.append(nn.Tanh())
        self.conv_block = nn.Sequential(*layers)


class SubPixelConvolutionalBlock(nn.Module):
    """conv -> PixelShuffle -> PReLU  (models/SRGAN.py:72-103)"""

    def __init__(self, kernel_size=3, n_channels=64, scaling_factor=2):
        super().__init__()
        self.conv = nn.Conv2d(n_channels, n_channels * scaling_factor ** 2, kernel_size, padding=kernel_size // 2)
        self.pixel_shuffle = nn.PixelShuffle(scaling_factor)
        self.prelu = nn.PReLU()


class ResidualBlock(nn.Module):
    """models/SRGAN.py:105-137"""

    def __init__(self, kernel_size=3, n_channels=64, inner_channels=None):
        super().__init__()
        inner = n_channels if inner_channels is None else inner_channels
        self.conv_block1 = ConvolutionalBlock(n_channels, inner, kernel_size, batch_norm=True, activation='PReLu')
        self.conv_block2 = ConvolutionalBlock(inner, n_channels, kernel_size, batch_norm=True, activation=None)


class Generator(nn.Module):
    """SRResNet parameter tree (models/SRGAN.py:139-199)"""

    def __init__(self, large_kernel_size=9, small_kernel_size=3, n_channels=64, n_blocks=16, scaling_factor=4, filter_cfgs=None):
        super().__init__()
        if (large_kernel_size, small_kernel_size, int(scaling_factor)) != (9, 3, 4):
            raise NotImplementedError('the MI355X path implements the configuration the reference trains: k9 / k3 / x4')
        self.conv_block1 = ConvolutionalBlock(3, n_channels, large_kernel_size, batch_norm=False, activation='PReLu')
        self.residual_blocks = nn.Sequential(*[ResidualBlock(small_kernel_size, n_channels,
                                                             None if filter_cfgs is None else int(filter_cfgs[i]))
                                               for i in range(n_blocks)])
        self.conv_block2 = ConvolutionalBlock(n_channels, n_channels, small_kernel_size, batch_norm=True, activation=None)
        self.subpixel_convolutional_blocks = nn.Sequential(*[SubPixelConvolutionalBlock(small_kernel_size, n_channels, 2)
                                                             for _ in range(int(math.log2(scaling_factor)))])
        self.conv_block3 = ConvolutionalBlock(n_channels, 3, large_kernel_size, batch_norm=False, activation='Tanh')

    def forward(self, lr_imgs):
        raise GccError('Generator owns parameters only; run it through SRGAN')


def _disc_tree(self, kernel_size, n_channels, n_blocks, mask, threshold):
    in_channels, blocks = 3, []
    for i in range(n_blocks):
        out_channels = (n_channels if i == 0 else in_channels * 2) if i % 2 == 0 else in_channels
        blocks.append(ConvolutionalBlock(in_channels, out_channels, kernel_size, stride=1 if i % 2 == 0 else 2,
                                         batch_norm=i != 0, activation='LeakyReLu', mask=mask, threshold=threshold))
        in_channels = out_channels
    self.conv_blocks = nn.Sequential(*blocks)
    self.adaptive_pool = nn.AdaptiveAvgPool2d((1, 1))
    self.fc1 = nn.Linear(out_channels, 1)


class Discriminator(nn.Module):
    """models/SRGAN.py:201-247"""

    def __init__(self, kernel_size=3, n_channels=64, n_blocks=4):
        super().__init__()
        _disc_tree(self, kernel_size, n_channels, n_blocks, False, 0.5)


class MaskDiscriminator(nn.Module):
    """models/SRGAN.py:249-297"""

    def __init__(self, kernel_size=3, n_channels=64, n_blocks=4, threshold=0.5):
        super().__init__()
        _disc_tree(self, kernel_size, n_channels, n_blocks, True, threshold)


class TruncatedVGG19(nn.Module):
    """vgg19.features up to the j-th conv (+ReLU) before the i-th max pool (models/GANLoss.py:95-145); i=5, j=4 is
    features[:36].  ``widths`` replaces the channel plan (tests use a narrow stand-in, as the golden fixture does)."""

    def __init__(self, i=5, j=4, widths=None):
        super().__init__()
        layers, cin, pools, convs = [], 3, 0, 0
        for c in (widths or VGG19_CFG):
            if c == 'M':
                layers.append(nn.MaxPool2d(2, 2))
                pools, convs = pools + 1, 0
            else:
                layers += [nn.Conv2d(cin, c, 3, padding=1), nn.ReLU(inplace=True)]
                cin, convs = c, convs + 1
                if pools == i - 1 and convs == j:
                    break
        assert pools == i - 1 and convs == j, 'One or both of i=%d and j=%d are not valid choices for the VGG19!' % (i, j)
        self.truncated_vgg19 = nn.Sequential(*layers)

    def load_torchvision(self, path):
        sd = torch.load(path, map_location='cpu')
        own = self.truncated_vgg19.state_dict()
        self.truncated_vgg19.load_state_dict({k: sd['features.' + k] for k in own})


class SRGAN(TeacherStreamMixin, nn.Module):

    def __init__(self, opt, filter_cfgs=None, channel_cfgs=None, vgg_widths=None):
        super().__init__()
        self.opt = opt
        if len(opt.gpu_ids) == 0 or not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU and gpu_ids >= 0')
        self.device = gdist.local_device(opt)
        ops.lib()
        self.filter_cfgs, self.channel_cfgs = filter_cfgs, channel_cfgs
        self.current_epoch = 0
        self.optimizers = []
        self.teacher_model = None
        self.visual_names = ['real_lr', 'fake_hr', 'real_hr']
        self.loss_names = ['content'] if opt.generator_only else ['G_GAN', 'D_real', 'D_fake', 'content', 'perceptual']
        self.generator_extract_layers = ['residual_blocks.3', 'residual_blocks.7', 'residual_blocks.11', 'residual_blocks.15']
        self.discriminator_extract_layers = ['conv_blocks.1', 'conv_blocks.3']
        dev = self.device
        masked = bool(opt.darts_discriminator)
        self.distill = bool(opt.online_distillation or opt.normal_distillation)

        self.netG = Generator(n_channels=opt.ngf, filter_cfgs=filter_cfgs)
        self.truncated_vgg19 = TruncatedVGG19(i=5, j=4, widths=vgg_widths)
        self.truncated_vgg19.eval()
        path = os.environ.get('GCC_VGG19_WEIGHTS')
        if path:
            self.truncated_vgg19.load_torchvision(path)
        elif os.environ.get('GCC_VGG19_RANDOM') == '1':
            g = torch.Generator().manual_seed(19)
            with torch.no_grad():
                for m in self.truncated_vgg19.modules():
                    if isinstance(m, nn.Conv2d):
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / (m.weight.shape[1] * 9)) ** 0.5)
                        m.bias.zero_()
        else:
            raise GccError('SRGAN needs the torchvision VGG19 ImageNet weights: set GCC_VGG19_WEIGHTS to a vgg19 state_dict '
                           'file (the reference downloads it), or GCC_VGG19_RANDOM=1 for seeded random weights')
        self.transform_convs = []
        if self.distill:
            self.transform_convs = [nn.Conv2d(opt.ngf, opt.teacher_ngf, 1, 1, 0, bias=False).to(dev) for _ in range(4)]
            for t in self.transform_convs:
                gdist.broadcast_module(t)          # default-initialised from each rank's RNG: replicas must start equal
        if masked:
            self.loss_names += ['D_arch_diff', 'D_arch', 'teacher_D_arch_diff']
            self.netD = MaskDiscriminator(n_channels=opt.ndf, threshold=opt.threshold)
        else:
            self.netD = Discriminator(n_channels=opt.ndf)
        self.init_net()

        # ---- optimizers (:327-372): with distillation only Conv / BatchNorm / Linear parameters (no PReLU)
        g_params, g_l1 = [], []
        for m in self.netG.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)) or (isinstance(m, nn.PReLU) and not self.distill):
                for p in m.parameters():
                    g_params.append(p)
                    if isinstance(m, nn.Conv2d) and p.dim() == 4 and opt.lambda_weight > 0.0:
                        g_l1.append(opt.lambda_weight)
                    elif isinstance(m, nn.BatchNorm2d) and p is m.weight and opt.lambda_scale > 0.0 and not opt.lambda_weight > 0.0:
                        g_l1.append(opt.lambda_scale)
                    else:
                        g_l1.append(0.0)
        g_params += [t.weight for t in self.transform_convs]
        g_l1 += [0.0] * len(self.transform_convs)
        self._sparsity = g_l1
        self.optimizer_G = HipAdam(g_params, lr=opt.lr, l1=[0.0] * len(g_params))
        # L1_sparsity() is only applied by optimize_content_parameters (:505-513): a second plan with the penalties
        self._content_plan = None
        w_params, a_params = [], []
        for m in self.netD.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d, nn.Linear)):
                w_params += list(m.parameters())
            elif isinstance(m, DifferentiableOP):
                a_params += list(m.parameters())
        self.optimizer_D = HipAdam(w_params, lr=opt.lr)
        if masked:
            self.optimizer_arch = HipAdam(a_params, lr=opt.arch_lr)
            if opt.arch_lr_step:
                self.optimizers.append(self.optimizer_arch)
        if opt.generator_only:
            self.optimizers.clear()
        self.optimizers += [self.optimizer_G, self.optimizer_D]
        self.schedulers = [util.get_scheduler(o, opt) for o in self.optimizers]

        # ---- engines
        for m in self.truncated_vgg19.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
                m.requires_grad_(False)
        self.G = engine.SRResNetEngine(self.netG, dev, train_prelu=not self.distill)
        self.D = engine.SRDiscriminatorEngine(self.netD, masked, opt.threshold, dev)
        self.V = engine.VGGEngine(self.truncated_vgg19, dev)
        self.V.repack()
        self.T = [engine.ConvOp(t.weight, None, 1, 1, 0, False) for t in self.transform_convs]
        self.refresh_weights()
        s = torch.tensor([1.0 / (2.0 * sd) for sd in IMAGENET_STD], dtype=torch.float32, device=dev)
        t = torch.tensor([(0.5 - m) / sd for m, sd in zip(IMAGENET_MEAN, IMAGENET_STD)], dtype=torch.float32, device=dev)
        self._norm_scale, self._norm_shift, self._zero3 = s, t, torch.zeros(3, dtype=torch.float32, device=dev)
        self._lossvec = torch.zeros(32, dtype=torch.float32, device=dev)
        self._slot = {n: i for i, n in enumerate(
            ['G_GAN', 'D_real', 'D_fake', 'mse_content', 'perceptual', 'L1', 'D_arch_fake', 'D_arch_fake_real', 'D_arch_real',
             'D_arch_diff', 'D_arch', 'teacher_D_arch_diff', 'arch_c_fr', 'arch_c_f', 'scratch0', 'scratch1', 'scratch2'])}
        self._dist_out = torch.zeros((6, 2), dtype=torch.float32, device=dev)
        self._bufs = {}
        self._nchw = {}
        self._ema_started = False
        self._world = gdist.world_size()

    # ---------------------------------------------------------------------------------------
    def _l(self, name):
        i = self._slot[name]
        return self._lossvec[i:i + 1]

    def refresh_weights(self):
        self.G.repack()
        self.D.repack()
        for t in self.T:
            t.repack()

    def init_net(self):
        for net in (self.netG, self.netD, self.truncated_vgg19):
            net.to(self.device)
        for m in self.netD.modules():
            if isinstance(m, DifferentiableOP):
                m.threshold = m.threshold.to(self.device)
        util.init_weights(self.netG, init_type='normal', init_gain=0.02)
        util.init_weights(self.netD, init_type='normal', init_gain=0.02)
        gdist.broadcast_module(self.netG)
        gdist.broadcast_module(self.netD)

    # ---------------------------------------------------------------------------------------
    def _buf(self, key, N, C, H, W):
        key = (key, N, C, H, W)
        if key not in self._bufs:
            self._bufs[key] = ops.new_act(N, C, H, W, self.device)
        return self._bufs[key]

    def _dws(self, i, N, C, HW):
        key = ('ws', i, N, C, HW)
        need = ops.distill_workspace_bytes(N, C, HW)       # depends on the weight-gradient split plan (tuning options)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = self._bufs[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    def set_input(self, input):
        self.input = input
        self._note_input(input)
        self.real_lr = input['lr'].to(self.device, torch.float32).contiguous()
        self._real_hr_nchw = input['hr'].to(self.device, torch.float32).contiguous()
        self.image_paths = [input.get('lr_names'), input.get('hr_names')]
        N, _, h, w = self.real_lr.shape
        self._lr = self._buf('lr', N, 3, h, w)
        self._hr = self._buf('hr', N, 3, 4 * h, 4 * w)
        self._hr_n = self._buf('hr_n', N, 3, 4 * h, 4 * w)
        self._fake_n = self._buf('fake_n', N, 3, 4 * h, 4 * w)
        ops.nchw_to_nhwc(self.real_lr, self._lr)
        ops.nchw_to_nhwc(self._real_hr_nchw, self._hr)
        self._hr_is_norm = False
        self._nchw = {}

    def forward(self):
        N, _, h, w = self._lr.shape
        c = self.G._ctx(N, h, w)
        ops.nhwc_copy(self._lr, 0, c.x_in, 0, 3)
        self._gctx = self.G.forward(c, train=self.netG.training)
        self._fake = self._gctx.out
        self._fake_is_norm = False
        self._nchw = {}

    def _normalise(self):
        """convert_image(., '[-1, 1]', 'imagenet-norm') of real_hr and fake_hr, as backward_G / get_D_arch_diff do"""
        ops.bnact_fwd(self._hr, self._hr_n, scale=self._norm_scale, shift=self._norm_shift)
        ops.bnact_fwd(self._fake, self._fake_n, scale=self._norm_scale, shift=self._norm_shift)
        self._hr_is_norm = self._fake_is_norm = True
        self._nchw = {}

    @property
    def fake_hr(self):
        if 'fake' not in self._nchw:
            self._nchw['fake'] = ops.nhwc_to_nchw(self._fake_n if self._fake_is_norm else self._fake, 3)
        return self._nchw['fake']

    @property
    def real_hr(self):
        if 'real' not in self._nchw:
            self._nchw['real'] = ops.nhwc_to_nchw(self._hr_n if self._hr_is_norm else self._hr, 3)
        return self._nchw['real']

    @property
    def Tfake_hr(self):
        return self.teacher_model.fake_hr

    def _d_forward(self, tag, img):
        N, _, H, W = img.shape
        ctx = self.D.new_ctx(N, H, W, tag)
        ops.nhwc_copy(img, 0, ctx.x_in, 0, 3)
        self.D.forward(ctx, train=True)
        return ctx

    def _allreduce(self, optimizer):
        gdist.all_reduce_grads(optimizer)

    # -- generator (:446-480) ---------------------------------------------------------------------------------------
    def backward_G(self, ts=None):
        opt, mode, gc = self.opt, self.opt.gan_mode, self._gctx
        N, _, H, W = self._fake.shape
        g_img = self._buf('g_img', N, 3, H, W)               # dL/d(fake_hr in [-1, 1]) from the MSE content term
        ops.mse_loss(self._fake, self._hr, self._l('mse_content'), weight=opt.lambda_SR_content, da=g_img)
        self._normalise()
        g_n = self._buf('g_n', N, 3, H, W)                   # dL/d(normalised fake_hr)
        def perceptual():
            # perceptual term: MSE between VGG feature maps of the normalised fake and real images
            vf, vr = self.V.new_ctx(N, H, W, 'fake'), self.V.new_ctx(N, H, W, 'real')
            ops.nhwc_copy(self._fake_n, 0, vf.x_in, 0, 3)
            ops.nhwc_copy(self._hr_n, 0, vr.x_in, 0, 3)
            ff, fr = self.V.forward(vf), self.V.forward(vr)
            g_ff = self._buf('g_ff', *ff.shape)
            ops.mse_loss(ff, fr, self._l('perceptual'), weight=opt.lambda_SR_perceptual, da=g_ff)
            return self.V.backward(vf, g_ff)

        # the VGG chain (two forwards, one backward; frozen weights) and the discriminator's pass over the fake only meet in
        # dL/d(fake_n): the VGG chain runs on the auxiliary stream beside the discriminator's (GCC_SR_FORK; the online teacher,
        # already on a stream of its own, keeps it in line); the two gradients are added in the reference's order
        aux = self._aux_stream() if (SR_FORK and not getattr(self, '_no_fork', False)) else False
        if aux:
            ops.wait_stream(aux, ops.current_stream())
            with ops.on_stream(aux):
                gv = perceptual()
        cg = self._d_forward('g_fake', self._fake_n)
        self._dctx_last = cg
        gp = self.D.grad_pred_buffer(cg)
        ops.gan_loss(mode, cg.pred, True, True, self._l('G_GAN'), dpred=gp, grad_weight=opt.lambda_SR_adversarial)
        dx = self.D.backward(cg, wgrad=False, need_dx=True)
        ops.nhwc_copy(dx, 0, g_n, 0, 3)
        if aux:
            ops.wait_stream(ops.current_stream(), aux)
        else:
            gv = perceptual()
        ops.nhwc_add(gv, 0, g_n, 0, 3)
        g_feat = None
        if self.distill:
            T = self.teacher_model
            self._join(ts)                                       # first read of the teacher's state
            ct = T._d_forward('on_student', self._fake_n)
            feats = self.G.features(gc) + T.D.features(ct)
            tf, dtf = [], []
            for i in range(4):
                f = feats[i]
                buf = self._buf(('tf', i), N, self.T[i].rows, f.shape[2], f.shape[3])
                self.T[i].forward(f, buf)
                tf.append(buf)
            tf += feats[4:]
            for i in range(6):
                dtf.append(self._buf(('dtf', i), N, tf[i].shape[1], tf[i].shape[2], tf[i].shape[3]))
                ws = self._dws(i, N, tf[i].shape[1], tf[i].shape[2] * tf[i].shape[3])
                t = self.target_distillation_features[i]
                ops.distill_fwd(tf[i], t, self._dist_out[i], ws)
                ops.distill_bwd(tf[i], t, opt.lambda_gram, opt.lambda_content, dtf[i], ws)
            g_feat = []
            for i in range(4):
                self.T[i].backward_weight(feats[i], dtf[i])
                gbuf = self._buf(('gf', i), N, feats[i].shape[1], feats[i].shape[2], feats[i].shape[3])
                self.T[i].backward_data(dtf[i], gbuf)
                g_feat.append(gbuf)
            ops.SideStream.get(self.device).join()
            dx2 = T.D.backward(ct, has_pred_grad=False, g_feat=[dtf[4], dtf[5]], wgrad=False, need_dx=True)
            ops.nhwc_add(dx2, 0, g_n, 0, 3)
            tmp = self._buf('l1', N, 3, H, W)
            ops.l1_loss(self._fake_n, T._fake_n, self._l('L1'), weight=opt.lambda_L1, da=tmp)
            ops.nhwc_add(tmp, 0, g_n, 0, 3)
            self._mark_teacher_free()
        # chain rule through the normalisation: d(fake_n)/d(fake) = 1 / (2 std_c)
        ops.bnact_fwd(g_n, g_n, scale=self._norm_scale, shift=self._zero3)
        ops.nhwc_copy(g_img, 0, gc.g_out, 0, 3)
        ops.nhwc_add(g_n, 0, gc.g_out, 0, 3)
        self.G.backward(gc, g_feat=g_feat, wgrad=True)

    # -- discriminator (:378-388): on the normalised images backward_G left behind, real first ----------------------------
    def backward_D(self):
        mode = self.opt.gan_mode
        cr = self._d_forward('d_real', self._hr_n)
        cf = self._d_forward('d_fake', self._fake_n)
        self._dctx_last = cf
        gp = self.D.grad_pred_buffer(cr)
        ops.gan_loss(mode, cr.pred, True, True, self._l('D_real'), dpred=gp)
        self.D.backward(cr, wgrad=True, need_dx=False)
        ops.gan_loss(mode, cf.pred, False, True, self._l('D_fake'), dpred=gp)
        self.D.backward(cf, wgrad=True, need_dx=False)

    def optimize_parameters(self):
        ts = None
        if self.opt.online_distillation:
            T = self.teacher_model

            T._no_fork = True            # the online teacher already runs on a stream of its own

            def teacher_step():
                T.set_input(self.input)
                prev = engine.OVERLAP_WGRAD
                if TEACHER_CHAIN_WGRAD:          # the teacher's weight gradients on its own stream: one HIP stream fewer
                    engine.OVERLAP_WGRAD = False
                try:
                    T.optimize_parameters()
                finally:
                    engine.OVERLAP_WGRAD = prev
                self.target_distillation_features = T.get_distillation_features()     # read after _join
            ts = self._run_teacher(teacher_step)
        self.forward()
        self.optimizer_G.zero_grad()
        self.backward_G(ts)
        self._allreduce(self.optimizer_G)
        self.optimizer_G.step()
        self.G.repack()
        for t in self.T:
            t.repack()
        self.optimizer_D.zero_grad()
        self.backward_D()
        self._allreduce(self.optimizer_D)
        self.optimizer_D.step()
        self.D.repack()

    def optimize_content_parameters(self):
        """generator-only pre-training step (:505-513): plain MSE + L1_sparsity()"""
        self.forward()
        self.optimizer_G.zero_grad()
        gc = self._gctx
        ops.mse_loss(self._fake, self._hr, self._l('mse_content'), weight=1.0, da=gc.g_out)
        self.G.backward(gc, wgrad=True)
        self._allreduce(self.optimizer_G)
        if any(v != 0.0 for v in self._sparsity):
            if self._content_plan is None:
                og = self.optimizer_G
                self._content_plan = ops.AdamPlan(og.plan.params, og.plan.grads, self.device, l1=self._sparsity)
                self._content_plan.m, self._content_plan.v = og.plan.m, og.plan.v
                self._content_plan._build()
            og = self.optimizer_G
            self._content_plan.step_count = og.plan.step_count
            g = og.param_groups[0]
            self._content_plan.step(g['lr'], g['betas'], g['eps'])
            og.plan.step_count = self._content_plan.step_count
        else:
            self.optimizer_G.step()
        self.G.repack()

    # -- architecture step (:390-424, 495-503) -------------------------------------------------------------------------
    def get_D_arch_diff(self, isTeacher=False):
        mode = self.opt.gan_mode
        self._normalise()
        cf = self._d_forward('a_fake', self._fake_n)
        cr = self._d_forward('a_real', self._hr_n)
        ops.gan_loss(mode, cf.pred, False, True, self._l('D_arch_fake'))
        ops.gan_loss(mode, cf.pred, True, False, self._l('D_arch_fake_real'))
        ops.gan_loss(mode, cr.pred, True, True, self._l('D_arch_real'))
        out = self._l('teacher_D_arch_diff' if isTeacher else 'D_arch_diff')
        if isTeacher and self._ema_started:
            b = float(self.opt.ema_beta)
            ops.scalar_op(1, self._l('D_arch_fake_real'), self._l('D_arch_fake'), out, c=out, k0=b, k1=1.0 - b)
        else:
            ops.scalar_op(0, self._l('D_arch_fake_real'), self._l('D_arch_fake'), out)
        self._ema_started = True
        return cf, cr

    def backward_D_arch(self, ts=None):
        T, mode = self.teacher_model, self.opt.gan_mode
        if not ts:
            T.get_D_arch_diff(isTeacher=True)
        cf, cr = self.get_D_arch_diff(isTeacher=False)
        self._join(ts)
        ops.scalar_op(2, T._l('teacher_D_arch_diff'), T._l('teacher_D_arch_diff'), self._l('teacher_D_arch_diff'), k0=0.0)
        self._mark_teacher_free()
        ops.arch_coeffs(self._l('D_arch_fake_real'), self._l('D_arch_fake'), self._l('D_arch_real'),
                        self._l('teacher_D_arch_diff'), self._l('D_arch'), self._l('arch_c_fr'), self._l('arch_c_f'), weight=1.0)
        gp = self.D.grad_pred_buffer(cf)
        ops.gan_loss(mode, cf.pred, True, False, self._l('scratch0'), dpred=gp, weight_dev=self._l('arch_c_fr'))
        ops.gan_loss(mode, cf.pred, False, True, self._l('scratch1'), dpred=gp, weight_dev=self._l('arch_c_f'),
                     dpred_accumulate=True)
        self.D.backward(cf, wgrad=False, agrad=True, need_dx=False)
        ops.gan_loss(mode, cr.pred, True, True, self._l('scratch2'), dpred=gp)
        self.D.backward(cr, wgrad=False, agrad=True, need_dx=False)

    def optimizer_netD_arch(self):
        T = self.teacher_model

        def teacher_part():
            T.set_input(self.input)
            T.forward()
            if self._teacher_stream():
                T.get_D_arch_diff(isTeacher=True)
        ts = self._run_teacher(teacher_part)
        self.forward()
        self.optimizer_arch.zero_grad()
        self.backward_D_arch(ts)
        self._allreduce(self.optimizer_arch)
        self.optimizer_arch.step()

    def clipping_mask_alpha(self):
        for m in self.netD.modules():
            if isinstance(m, DifferentiableOP):
                m.clip_alpha()

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
        self.current_epoch = epoch
        print('learning rate = %.7f' % self.optimizers[0].param_groups[0]['lr'])

    def set_requires_grad(self, nets, requires_grad=False):
        for net in (nets if isinstance(nets, list) else [nets]):
            if net is not None:
                for p in net.parameters():
                    p.requires_grad = requires_grad

    def save_models(self, epoch, save_dir, fid=None, isbest=False, direction='AtoB'):
        if gdist.rank() != 0:
            return
        util.mkdirs(save_dir)
        ckpt = {'G': _portable(self.netG.state_dict()), 'D': _portable(self.netD.state_dict()), 'epoch': epoch,
                'cfg': (self.filter_cfgs, self.channel_cfgs), 'psnr': fid}
        name = 'model_best_%s.pth' % direction if isbest else 'model_%d.pth' % epoch
        torch.save(ckpt, os.path.join(save_dir, name))

    def load_models(self, load_path, load_discriminator=True):
        ckpt = torch.load(load_path, map_location='cpu')
        self.netG.load_state_dict(ckpt['G'])
        if load_discriminator:
            self.netD.load_state_dict(ckpt['D'])
        self.refresh_weights()
        print('loading the model from %s' % load_path)
        return ckpt['psnr'], float('inf')

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
        v = self._lossvec.cpu()
        d = self._dist_out.cpu()
        s = self._slot
        ret = OrderedDict()
        for name in self.loss_names:
            if name == 'content':
                val = self.opt.lambda_content * float(d[:, 1].sum()) if self.distill else float(v[s['mse_content']])
            elif name == 'gram':
                val = self.opt.lambda_gram * float(d[:, 0].sum())
            elif name == 'G_GAN':
                val = self.opt.lambda_SR_adversarial * float(v[s['G_GAN']])
            else:
                val = float(v[s[name]])
            ret[name] = val
        if self._world > 1:
            ret = gdist.mean_dict(ret, self.device)
        return ret

    def get_current_psnr(self):
        """PSNR on the luminance channel with a 4-pixel border cropped (convert_image 'y-channel', :653-657;
        skimage.metrics.peak_signal_noise_ratio with data_range 255 = 10 log10(255^2 / mse))"""
        fake, real = self.fake_hr.float().contiguous(), self.real_hr.float().contiguous()
        N, _, H, W = fake.shape
        L = ops.lib()
        sse = torch.zeros(1, dtype=torch.float64, device=fake.device)
        ws = torch.empty(L.gcc_psnr_workspace(), dtype=torch.uint8, device=fake.device)
        check(L.gcc_psnr_y_sse(fake.data_ptr(), real.data_ptr(), N, H, W, sse.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                               ops.stream()), 'gcc_psnr_y_sse')
        mse = float(sse.item()) / (N * (H - 8) * (W - 8))
        return 10.0 * math.log10(255.0 ** 2 / mse) if mse > 0 else float('inf')

    def get_current_ssim(self):
        """models/SRGAN.py:659-661: skimage's structural_similarity(real_y, fake_y, data_range=255.) on the same cropped
        luminance images (7 x 7 uniform window, sample covariance, K1 .01 / K2 .03, 3-pixel border of the SSIM map dropped);
        skimage is un-pinned and absent from the build image: restated from its published definition, parity unpinned"""
        fake, real = self.fake_hr.float().contiguous(), self.real_hr.float().contiguous()
        N, _, H, W = fake.shape
        L = ops.lib()
        acc = torch.zeros(1, dtype=torch.float64, device=fake.device)
        ws = torch.empty(L.gcc_psnr_workspace(), dtype=torch.uint8, device=fake.device)
        check(L.gcc_ssim_y_sum(fake.data_ptr(), real.data_ptr(), N, H, W, acc.data_ptr(), 0, ws.data_ptr(), ws.numel(),
                               ops.stream()), 'gcc_ssim_y_sum')
        return float(acc.item()) / (N * (H - 14) * (W - 14))

    def init_distillation(self):
        if self.distill:
            if self.opt.lambda_content > 0.0:
                self.loss_names.append('content')
            if self.opt.lambda_gram > 0.0:
                self.loss_names.append('gram')
            if self.opt.lambda_L1 > 0.0:
                self.loss_names.append('L1')
            self.visual_names.append('Tfake_hr')

    def get_distillation_features(self):
        """4 generator features (outputs of residual blocks 3/7/11/15) + the 2 discriminator features of the last D call"""
        return self.G.features(self._gctx) + self.D.features(self._dctx_last)

    def get_cfg(self):
        return self.filter_cfgs, self.channel_cfgs

    # -- pruning (models/SRGAN.py:703-830): per-block inner widths from BatchNorm scales or filter norms ----------------
    def max_min_bn_scale(self):
        from ..utils import prune_util
        return prune_util.srgan_max_min_bn_scale(self.netG)

    def max_min_conv_norm(self):
        from ..utils import prune_util
        return prune_util.srgan_max_min_conv_norm(self.netG)

    def _pruned(self, threshold, scale, lottery_path):
        from ..utils import prune_util
        cfgs, _ = prune_util.srgan_prune_cfg(self.netG, threshold, scale)
        pruned = SRGAN(self.opt, filter_cfgs=cfgs)
        if lottery_path is not None:
            # the reference calls pruned_model.lottery_theory, which models/SRGAN.py does not define (:795, :828)
            raise AttributeError("'SRGAN' object has no attribute 'lottery_theory'")
        return pruned

    def norm_prune(self, threshold, lottery_path=None):
        return self._pruned(threshold, False, lottery_path)

    def scale_prune(self, threshold, lottery_path=None):
        return self._pruned(threshold, True, lottery_path)

    def prune(self, threshold, lottery_path=None):
        if self.opt.scale_prune:
            return self.scale_prune(threshold, lottery_path)
        elif self.opt.norm_prune:
            return self.norm_prune(threshold, lottery_path)
        raise NotImplementedError('only scale and norm pruning are supported!!!')
