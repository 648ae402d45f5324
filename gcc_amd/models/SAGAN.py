"""SAGAN GCC model on MI355X -- the reference's ``models/SAGAN.py`` surface over the HIP engine.

Kept name-for-name: SpectralNorm / Self_Attn / Generator / Discriminator / MaskDiscriminator / SAGANModel, their
constructor signatures, the state_dict keys (``l1.0.module.weight_bar`` / ``weight_u`` / ``weight_v`` ...,
``attn1.gamma`` ...), ``loss_names`` / ``visual_names``, the checkpoint dict layout.

Reference behaviour that is reproduced on purpose (SURVEY.md section 8 hazards):
  * the power iteration of every spectrally normalised conv runs on EVERY forward of the net, eval included, and moves
    the stored u, v (models/SAGAN.py:25-38, 66-68);
  * ``set_requires_grad(netD, True)`` also switches on the discriminator's u, v: they get gradients through sigma and
    Adam updates in each D step (:513); the generator's never do;
  * the distillation / masked-D optimizer lists hold SpectralNorm and Self_Attn containers AND their child convs, so
    those tensors are updated twice per step (H5; HipAdam ``dup``);
  * MaskDiscriminator ignores --threshold (H6: the constructor call passes none, so tau = 0.5);
  * with distillation the terms are accumulated in place into the tensor that is also ``loss_G_GAN`` (:465-488): the
    reported G_GAN is the whole generator loss;
  * Adam betas (0, 0.9); the discriminator's learning rate is 4x; only the arch scheduler steps.
"""
import copy
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn import Parameter

from .. import dist as gdist
from .. import engine, ops
from .._lib import GccError
from ..utils import util
from .DifferentiableOp import DifferentiableOP
from .Pix2Pix import HipAdam, _portable
from ._streams import TeacherStreamMixin


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


# GCC_SAGAN_CHAIN_WGRAD (default 1): the weight-gradient launches stay on the stream of the chain that needs them instead of a side
# stream: at 64 x 64 every kernel is a few microseconds and the two event operations of a side-stream fork cost more than the
# overlap returns (eager 10.6 -> 8.8 ms, replayed 9.3 -> 8.4: profiles/r4ak_chain_wgrad.txt; SRGAN measured the other way round)
CHAIN_WGRAD = os.environ.get('GCC_SAGAN_CHAIN_WGRAD', '1') != '0'
# GCC_SAGAN_FORK=1: backward_G's distillation block on the auxiliary stream beside the discriminator's pass -- same bits, measured
# SLOWER (replayed 8.42 -> 9.0 ms, profiles/r4an_sagan_fork.txt: the chains are too short to pay for a third stream): off
G_FORK = os.environ.get('GCC_SAGAN_FORK', '0') == '1'


class _ChainWgrad:
    def __enter__(self):
        self.prev = engine.OVERLAP_WGRAD
        if CHAIN_WGRAD:
            engine.OVERLAP_WGRAD = False
        return self

    def __exit__(self, *exc):
        engine.OVERLAP_WGRAD = self.prev
        return False


class SpectralNorm(nn.Module):
    """Parameter holder of the reference's wrapper (models/SAGAN.py:17-70): the wrapped conv loses ``weight`` and gains
    ``weight_u`` [rows], ``weight_v`` [cols*k*k] (requires_grad False) and ``weight_bar``; the arithmetic lives in
    gcc_amd.engine.SNConvOp."""

    def __init__(self, module, name='weight', power_iterations=1):
        super().__init__()
        if name != 'weight' or power_iterations != 1:
            raise NotImplementedError('the MI355X path implements the configuration the reference uses: weight, 1 iteration')
        self.module, self.name, self.power_iterations = module, name, power_iterations
        w = module.weight
        height = w.shape[0]
        width = w.numel() // height
        u = Parameter(l2normalize(w.data.new(height).normal_(0, 1)), requires_grad=False)
        v = Parameter(l2normalize(w.data.new(width).normal_(0, 1)), requires_grad=False)
        w_bar = Parameter(w.data)
        del module._parameters['weight']
        module.register_parameter('weight_u', u)
        module.register_parameter('weight_v', v)
        module.register_parameter('weight_bar', w_bar)

    def forward(self, *args):
        raise GccError('SpectralNorm owns parameters only; run the net through SAGANModel')


class Self_Attn(nn.Module):
    """parameter holder (models/SAGAN.py:72-104): gamma, query / key (in_dim // 8 channels) / value 1x1 convs"""

    def __init__(self, in_dim, activation=None):
        super().__init__()
        self.chanel_in, self.activation = in_dim, activation
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))


class Generator(nn.Module):
    """models/SAGAN.py:106-170 (image_size 64): l4 is registered before l1..l3, as in the reference"""

    def __init__(self, ngf=64, image_size=64, z_dim=128, filter_cfgs=None):
        super().__init__()
        if image_size != 64:
            raise NotImplementedError('the MI355X path implements image_size 64 (the only size the reference trains)')
        self.imsize = image_size
        w = [int(v) for v in filter_cfgs] if filter_cfgs is not None else [ngf * 8, ngf * 4, ngf * 2, ngf]
        cin = [z_dim] + w[:3]
        geom = ((4, 1, 0), (4, 2, 1), (4, 2, 1), (4, 2, 1))

        def layer(i):
            k, s, p = geom[i]
            return nn.Sequential(SpectralNorm(nn.ConvTranspose2d(cin[i], w[i], k, s, p)), nn.BatchNorm2d(w[i]), nn.ReLU())
        self.l4 = layer(3)
        self.l1, self.l2, self.l3 = layer(0), layer(1), layer(2)
        self.last = nn.Sequential(nn.ConvTranspose2d(w[3], 3, 4, 2, 1), nn.Tanh())
        self.attn1 = Self_Attn(w[2], 'relu')
        self.attn2 = Self_Attn(w[3], 'relu')

    def forward(self, z):
        raise GccError('Generator owns parameters only; run it through SAGANModel')


def _disc_tree(self, ndf, masked, threshold):
    w = [ndf, ndf * 2, ndf * 4, ndf * 8]
    cin = [3] + w[:3]

    def layer(i):
        mods = [SpectralNorm(nn.Conv2d(cin[i], w[i], 4, 2, 1))]
        if masked:
            mods.append(DifferentiableOP(w[i], threshold))
        mods.append(nn.LeakyReLU(0.1))
        return nn.Sequential(*mods)
    self.l4 = layer(3)
    self.l1, self.l2, self.l3 = layer(0), layer(1), layer(2)
    self.last = nn.Sequential(nn.Conv2d(w[3], 1, 4))
    self.attn1 = Self_Attn(ndf * 4, 'relu')
    self.attn2 = Self_Attn(ndf * 8, 'relu')


class Discriminator(nn.Module):
    """models/SAGAN.py:172-220"""

    def __init__(self, ndf=64, image_size=64):
        super().__init__()
        self.imsize = image_size
        _disc_tree(self, ndf, False, 0.5)


class MaskDiscriminator(nn.Module):
    """models/SAGAN.py:222-274: a DifferentiableOP between each spectrally normalised conv and its LeakyReLU"""

    def __init__(self, ndf=64, image_size=64, threshold=0.5):
        super().__init__()
        self.imsize = image_size
        _disc_tree(self, ndf, True, threshold)


def _is_dup(name):
    """SpectralNorm-wrapped convs and the attention convs are collected through their container and again as children"""
    return ('.module.' in name) or ('_conv.' in name)


class SAGANModel(TeacherStreamMixin, nn.Module):

    def __init__(self, opt, filter_cfgs=None, channel_cfgs=None):
        super().__init__()
        self.opt = opt
        if len(opt.gpu_ids) == 0 or not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU and gpu_ids >= 0')
        self.device = gdist.local_device(opt)
        ops.lib()
        self.filter_cfgs, self.channel_cfgs = filter_cfgs, channel_cfgs
        self.loss_names = ['G_GAN', 'D_real', 'D_fake']
        self.visual_names = ['fake_img', 'real_img']
        self.generator_extract_layers = ['l2', 'attn2']
        self.discriminator_extract_layers = ['l2', 'attn2']
        self.teacher_model = None
        self.optimizers = []
        dev = self.device
        masked = bool(opt.darts_discriminator)

        self.netG = Generator(ngf=opt.ngf, image_size=opt.crop_size, z_dim=opt.z_dim, filter_cfgs=filter_cfgs)
        self.distill = bool(opt.online_distillation or opt.normal_distillation)
        self.transform_convs = []
        if self.distill:
            t_w = [opt.teacher_ngf * 4, opt.teacher_ngf]
            s_w = [opt.ngf * 4, opt.ngf] if filter_cfgs is None else [filter_cfgs[1], filter_cfgs[3]]
            self.transform_convs = [nn.Conv2d(s, t, 1, 1, 0, bias=False).to(dev) for s, t in zip(s_w, t_w)]
            for t in self.transform_convs:
                gdist.broadcast_module(t)          # default-initialised from each rank's RNG: replicas must start equal
        if masked:
            self.loss_names += ['D_arch_diff', 'D_arch', 'teacher_D_arch_diff']
            self.netD = MaskDiscriminator(ndf=opt.ndf)            # H6: --threshold is not forwarded
        else:
            self.netD = Discriminator(ndf=opt.ndf)
        self.init_net()

        # ---- optimizers (models/SAGAN.py:302-354); u, v of the generator never get a gradient and are left out
        g_named = [(n, p) for n, p in self.netG.named_parameters() if not (n.endswith('_u') or n.endswith('_v'))]
        g_params = [p for _, p in g_named] + [t.weight for t in self.transform_convs]
        g_l1 = []
        for n, p in g_named:
            is_conv_w = n.endswith('.weight') and p.dim() == 4          # SN convs own no '.weight': L1_sparsity skips them
            is_bn_w = n.endswith('.weight') and p.dim() == 1
            if opt.lambda_weight > 0.0:
                g_l1.append(opt.lambda_weight if is_conv_w else 0.0)
            elif opt.lambda_scale > 0.0:
                g_l1.append(opt.lambda_scale if is_bn_w else 0.0)
            else:
                g_l1.append(0.0)
        g_l1 += [0.0] * len(self.transform_convs)
        g_dup = [p for n, p in g_named if _is_dup(n)] if self.distill else []
        self.optimizer_G = HipAdam(g_params, lr=opt.lr, betas=(0.0, 0.9), l1=g_l1, dup=g_dup)
        d_named = [(n, p) for n, p in self.netD.named_parameters() if not n.endswith('.alpha')]
        d_dup = [p for n, p in d_named if _is_dup(n)] if masked else []
        self.optimizer_D = HipAdam([p for _, p in d_named], lr=opt.lr * 4, betas=(0.0, 0.9), dup=d_dup)
        if masked:
            self.optimizer_arch = HipAdam([p for n, p in self.netD.named_parameters() if n.endswith('.alpha')], lr=opt.arch_lr)
            if opt.arch_lr_step:
                arch_opt = copy.deepcopy(opt)
                arch_opt.lr_policy = 'step'
                arch_opt.lr_decay_iters = 40
                self.arch_scheduler = util.get_scheduler(self.optimizer_arch, arch_opt)

        # ---- engines
        self.G = engine.SaganGeneratorEngine(self.netG, dev)
        self.D = engine.SaganDiscriminatorEngine(self.netD, masked, 0.5, dev)
        self.T = [engine.ConvOp(t.weight, None, 1, 1, 0, False) for t in self.transform_convs]
        self.refresh_weights()
        self._lossvec = torch.zeros(32, dtype=torch.float32, device=dev)
        self._slot = {n: i for i, n in enumerate(
            ['G_GAN', 'D_real', 'D_fake', 'L1', 'D_arch_fake', 'D_arch_fake_real', 'D_arch_real', 'D_arch_diff', 'D_arch',
             'teacher_D_arch_diff', 'arch_c_fr', 'arch_c_f', 'scratch0', 'scratch1', 'scratch2'])}
        self._dist_out = torch.zeros((4, 2), dtype=torch.float32, device=dev)
        self._bufs = {}
        self._nchw = None
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
        for net in (self.netG, self.netD):
            net.to(self.device)
            for m in net.modules():
                if isinstance(m, DifferentiableOP):
                    m.threshold = m.threshold.to(self.device)
            util.init_weights(net, init_type='normal', init_gain=0.02)
            gdist.broadcast_module(net)

    # ---------------------------------------------------------------------------------------
    def set_input(self, input):
        self.input = input
        self._note_input(input)
        self.z = input['z'].to(self.device, torch.float32).contiguous()
        self.real_img = input['real_img'].to(self.device, torch.float32).contiguous()
        self.image_paths = [input.get('img_path'), input.get('img_path')]
        N = self.z.shape[0]
        if getattr(self, '_real', None) is None or self._real.shape[0] != self.real_img.shape[0]:
            self._real = ops.new_act(self.real_img.shape[0], 3, 64, 64, self.device)
        ops.nchw_to_nhwc(self.real_img, self._real)
        c = self.G._ctx(N)
        ops.nchw_to_nhwc(self.z.reshape(N, -1, 1, 1), c.z)

    def forward(self):
        """fake_img = G(z)  (:365-368); every call advances the generator's power iteration"""
        self._gctx = self.G.forward(self.G._ctx(self.z.shape[0]), train=self.netG.training)
        self._fake = self._gctx.out
        self._nchw = None

    @property
    def fake_img(self):
        if self._nchw is None:
            self._nchw = ops.nhwc_to_nchw(self._fake, 3)
        return self._nchw

    @property
    def Tfake_img(self):
        return self.teacher_model.fake_img

    def _d_forward(self, tag, img):
        ctx = self.D.new_ctx(img.shape[0], tag)
        ops.nhwc_copy(img, 0, ctx.x_in, 0, 3)
        self.D.forward(ctx)
        return ctx

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

    def _allreduce(self, optimizer):
        gdist.all_reduce_grads(optimizer)

    # -- D step (:370-381): real first, then the detached fake; the two terms are summed without the 1/2 ----------
    def backward_D(self):
        mode = self.opt.gan_mode
        cr = self._d_forward('d_real', self._real)
        cf = self._d_forward('d_fake', self._fake)
        gp = self.D.grad_pred_buffer(cr)
        ops.gan_loss(mode, cr.pred, True, True, self._l('D_real'), dpred=gp)
        self.D.backward(cr, wgrad=True, need_dx=False)
        ops.gan_loss(mode, cf.pred, False, True, self._l('D_fake'), dpred=gp)
        self.D.backward(cf, wgrad=True, need_dx=False)

    # -- G step (:460-494) ------------------------------------------------------------------------------------
    def backward_G(self, ts=None):
        opt, mode, gc = self.opt, self.opt.gan_mode, self._gctx
        g_feat = None

        def own_d():
            cg = self._d_forward('g_fake', self._fake)
            self._dctx_g = cg
            ops.gan_loss(mode, cg.pred, True, False, self._l('G_GAN'), dpred=self.D.grad_pred_buffer(cg))
            return self.D.backward(cg, wgrad=False, need_dx=True)

        def distill_terms():
            T = self.teacher_model
            N = gc.N
            self._join(ts)                                       # first read of the teacher's state (on this chain's stream)
            ct = T._d_forward('on_student', self._fake)          # teacher D (frozen) on the student's fake: not detached
            feats = self.G.features(gc) + T.D.features(ct)
            tf, dtf = [], []
            for i in range(2):
                f = feats[i]
                buf = self._buf(('tf', i), N, self.T[i].rows, f.shape[2], f.shape[3])
                self.T[i].forward(f, buf)
                tf.append(buf)
            tf += feats[2:]
            for i in range(4):
                dtf.append(self._buf(('dtf', i), N, tf[i].shape[1], tf[i].shape[2], tf[i].shape[3]))
                ws = self._dws(i, N, tf[i].shape[1], tf[i].shape[2] * tf[i].shape[3])
                t = self.target_distillation_features[i]
                ops.distill_fwd(tf[i], t, self._dist_out[i], ws)
                ops.distill_bwd(tf[i], t, opt.lambda_gram, opt.lambda_content, dtf[i], ws)
            gf = []
            for i in range(2):
                self.T[i].backward_weight(feats[i], dtf[i])
                gbuf = self._buf(('gf', i), N, feats[i].shape[1], feats[i].shape[2], feats[i].shape[3])
                self.T[i].backward_data(dtf[i], gbuf)
                gf.append(gbuf)
            ops.SideStream.get(self.device).join()
            dx2 = T.D.backward(ct, has_pred_grad=False, g_feat=[dtf[2], dtf[3]], wgrad=False, need_dx=True)
            tmp = self._buf('l1', N, 3, 64, 64)
            ops.l1_loss(self._fake, T._fake, self._l('L1'), weight=opt.lambda_L1, da=tmp)
            return gf, dx2, tmp

        # the student discriminator's pass over the fake and the distillation block (teacher discriminator over the same fake,
        # transform convs, gram / content terms) only meet in dL/d(fake): the block runs on the auxiliary stream beside the pass
        # (GCC_SAGAN_FORK; the online teacher keeps everything in line); the gradients are added in the reference's order
        aux = self._aux_stream() if (self.distill and G_FORK and not getattr(self, '_no_fork', False)) else False
        if aux:
            ops.wait_stream(aux, ops.current_stream())
            with ops.on_stream(aux):
                g_feat, dx2, tmp = distill_terms()
        dx = own_d()
        ops.nhwc_copy(dx, 0, gc.g_out, 0, 3)
        if self.distill:
            if aux:
                ops.wait_stream(ops.current_stream(), aux)
            else:
                g_feat, dx2, tmp = distill_terms()
            ops.nhwc_add(dx2, 0, gc.g_out, 0, 3)
            ops.nhwc_add(tmp, 0, gc.g_out, 0, 3)
            self._mark_teacher_free()
        self.G.backward(gc, g_feat=g_feat, wgrad=True)

    # -- one iteration (:508-528) -----------------------------------------------------------------------------
    def optimize_parameters(self):
        with _ChainWgrad():
            return self._optimize_parameters()

    def _optimize_parameters(self):
        ts = None
        if self.opt.online_distillation:
            T = self.teacher_model

            T._no_fork = True            # the online teacher already runs on a stream of its own

            def teacher_step():
                T.set_input(self.input)
                T.optimize_parameters()
                self.target_distillation_features = T.get_distillation_features()     # read after _join
            ts = self._run_teacher(teacher_step)
        self.forward()
        self.optimizer_D.zero_grad()
        self.backward_D()
        self._allreduce(self.optimizer_D)
        self.optimizer_D.step()
        self.D.repack()
        self.optimizer_G.zero_grad()
        self.backward_G(ts)
        self._allreduce(self.optimizer_G)
        self.optimizer_G.step()          # L1_sparsity() (:496-506) is fused into the Adam kernel
        self.G.repack()
        for t in self.T:
            t.repack()

    # -- architecture step (:383-413, 530-538) -----------------------------------------------------------------
    def get_D_arch_diff(self, isTeacher=False):
        mode = self.opt.gan_mode
        cf = self._d_forward('a_fake', self._fake)
        cr = self._d_forward('a_real', self._real)
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
        # loss_D_arch = |d_S - d_T| + L_real + L_fake  (no 1/2 here, :388-389)
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
        with _ChainWgrad():
            return self._optimizer_netD_arch()

    def _optimizer_netD_arch(self):
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
        if self.opt.arch_lr_step and hasattr(self, 'arch_scheduler'):
            self.arch_scheduler.step()
        self.adaptive_ema_beta(epoch)
        print('learning rate = %.7f' % self.optimizer_G.param_groups[0]['lr'])

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
        v = self._lossvec.cpu()
        d = self._dist_out.cpu()
        s = self._slot
        gram = self.opt.lambda_gram * float(d[:, 0].sum())
        content = self.opt.lambda_content * float(d[:, 1].sum())
        ret = OrderedDict()
        for name in self.loss_names:
            if name == 'content':
                val = content
            elif name == 'gram':
                val = gram
            elif name == 'G_GAN' and self.distill:
                val = float(v[s['G_GAN']]) + gram + content + float(v[s['L1']])       # in-place accumulation, :465-488
            else:
                val = float(v[s[name]])
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
            if self.opt.lambda_L1 > 0.0:
                self.loss_names.append('L1')
            self.visual_names.append('Tfake_img')

    def get_distillation_features(self):
        """2 generator features ('l2', 'attn2') + the 2 discriminator features of the last D call of the iteration"""
        return self.G.features(self._gctx) + self.D.features(self._dctx_g)

    def get_cfg(self):
        return self.filter_cfgs, self.channel_cfgs

    # -- pruning (:661-740): BatchNorm-scale counts per generator layer ---------------------------------------------
    def max_min_bn_scale(self):
        un_max, mn = None, None
        for m in self.netG.modules():
            if isinstance(m, nn.BatchNorm2d):
                w = m.weight.detach().float().cpu()
                un_max = w.max() if un_max is None else torch.min(w.max(), un_max)
                mn = w.min() if mn is None else torch.min(w.min(), mn)
        return un_max, mn

    def scale_prune(self, threshold):
        if torch.is_tensor(threshold):
            threshold = threshold.detach().float().cpu()
        cfg = {'l1': 0, 'l2': 0, 'l3': 0, 'l4': 0}
        for name, m in self.netG.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                cfg[name.split('.')[0]] = int((m.weight.detach().float().cpu() > threshold).sum())
        return SAGANModel(self.opt, filter_cfgs=list(cfg.values()))

    def max_min_conv_norm(self):
        """models/SAGAN.py:719-721: an empty method in the reference (`pass`): kept for the surface, returns None as there"""
        return None

    def norm_prune(self, threshold, lottery_path=None):
        """models/SAGAN.py:752-754: an empty method in the reference (`pass`): `--norm_prune` on SAGAN yields None there too"""
        return None

    def prune(self, threshold, lottery_path=None):
        if self.opt.scale_prune:
            return self.scale_prune(threshold)
        if getattr(self.opt, 'norm_prune', False):
            return self.norm_prune(threshold, lottery_path)           # :699-700
        raise NotImplementedError('only scale and norm pruning are supported!!!')
