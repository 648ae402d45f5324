"""MobileCycleGAN GCC model on MI355X -- the reference's ``models/CycleGAN.py`` surface
(set_input / forward / optimize_parameters / optimizer_netD_arch / save_models / ...) over the HIP engine.

Kept name-for-name: class names and constructor signatures, the ``netG_A`` / ``netG_B`` / ``netD_A`` / ``netD_B``
module trees (state_dict keys equal the reference's), ``loss_names`` / ``visual_names``, optimizers, schedulers and
the checkpoint dict layout ('G_A', 'G_B', 'D_A', 'D_B', 'epoch', 'cfg', 'fid').

The step is written out explicitly in the order of models/CycleGAN.py:571-600:
    [teacher step] -> forward -> generators (identity, GAN, cycle, distillation) -> Adam(G)
                   -> discriminators on (real, pooled fake) -> Adam(D);      arch step: alpha gates of both D.
forward() runs each generator once per distinct input: the reference calls G_A(real_A) and G_B(real_B) twice only
to leave its forward hooks pointing at those passes; InstanceNorm has no state and dropout is 0, so the second pass
is the same arithmetic and autograd would add the same gradients.
Naming: netG_A maps domain A -> B and is judged by netD_A (on domain B); side 'A' below means that pair.
"""
import copy
import os
import random
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as gdist
from .. import engine, ops
from .._lib import GccError
from ..utils import util
from .DifferentiableOp import DifferentiableOP
from .Pix2Pix import HipAdam, MobileResnetGenerator, _patchgan_tree, _portable
from ._streams import TeacherStreamMixin

HEAVY_SPARSITY = ('model.1', 'model.4', 'model.19', 'model.22')      # models/CycleGAN.py:243, 548-569
# GCC_CYCLE_FORK=1: the two sides of the model (generator A -> B with its discriminator, generator B -> A with its) are
# independent chains of small kernels inside forward, backward_G, backward_D and the architecture step: side B runs on the
# auxiliary stream beside side A.  The host enqueues A then B as before, so everything that accumulates in launch order on the
# weight-gradient side stream (both generators' parameter gradients) keeps the reference's order: same bits.
CYCLE_FORK = int(os.environ.get('GCC_CYCLE_FORK', '2'))
# 2 (default): the online teacher's two sides fork as well, and the weight gradients stay on their chain's stream instead of a
# side stream each: four chains on four HIP streams = the device's four hardware queues (with side streams the same forks
# make eight streams and lose: profiles/r4ag_cyclegan_streams.txt -- eager 32.2 -> 23.3 ms, replayed 30.0 -> 24.7);
# 1: the student's sides only, weight gradients on side streams (six streams: 25.6 ms replayed)


class _ChainWgrad:
    """with CYCLE_FORK >= 2: weight-gradient launches stay on the stream of the chain that needs them (engine.OVERLAP_WGRAD
    off for the duration of the step; restored afterwards: other model families of the process keep their side streams)"""

    def __enter__(self):
        self.prev = engine.OVERLAP_WGRAD
        if CYCLE_FORK >= 2 and os.environ.get('GCC_CYCLE_CHAIN_WGRAD', '1') != '0':
            engine.OVERLAP_WGRAD = False
        return self

    def __exit__(self, *exc):
        engine.OVERLAP_WGRAD = self.prev
        return False


class NLayerDiscriminator(nn.Module):
    """InstanceNorm PatchGAN parameter tree (models/CycleGAN.py:139-177): convs (all biased) at model.0/2/5/8/11"""

    def __init__(self, input_nc=3, ndf=64, n_layers=3):
        super().__init__()
        _patchgan_tree(self, input_nc, ndf, n_layers, False, 0.5, norm='instance')


class MaskNLayerDiscriminator(nn.Module):
    """Selective-activation PatchGAN with BatchNorm (models/CycleGAN.py:179-222): convs at model.0/3/7/11/15, BN at
    4/8/12, gates (alpha) at 2/5/9/13"""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, threshold=0.5):
        super().__init__()
        _patchgan_tree(self, input_nc, ndf, n_layers, True, threshold)


class ImagePool:
    """utils/image_pool.py:5-54 on device buffers: the history holds NHWC bf16 images; the random draws are Python's
    ``random`` in the reference's order (uniform per image once full, randint on a swap).  What the draws decided goes to the
    device as (mode, slot) per image, carried by a launch that takes them by value, and one kernel moves the pixels
    (gcc_image_pool_query): the iteration's launch sequence does not depend on the draws, so it can be recorded and
    replayed (gcc_amd.replay) with that one argument patched per iteration."""

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.count = 0              # images held
        self.store = None           # [pool_size, 3, H, W] NHWC bf16
        self.sel = None             # device int32 [8][2]: the draws of the group of images in flight
        self._groups = []

    def _draw(self, N):
        """(mode, slot) per image: 0 pass through, 1 store in `slot` and pass through, 2 swap with `slot`"""
        sel = []
        for _ in range(N):
            if self.pool_size == 0:
                sel += [0, 0]
            elif self.count < self.pool_size:
                sel += [1, self.count]
                self.count += 1
            elif random.uniform(0, 1) > 0.5:
                sel += [2, random.randint(0, self.pool_size - 1)]
            else:
                sel += [0, 0]
        return sel

    def query(self, images, out):
        """images: NHWC bf16 batch view [N,3,H,W]; out: batch buffer of the same geometry that receives the answer.
        The draws of up to 8 images travel in one launch argument (16 ints): a larger batch goes group by group, draws in image
        order as in utils/image_pool.py:23-52 (any --batch_size works, as in the reference)."""
        N, _, H, W = images.shape
        if self.store is None:
            self.store = ops.new_act(max(self.pool_size, 1), 3, H, W, images.device)
            self.sel = torch.zeros(16, dtype=torch.int32, device=images.device)
        elif tuple(self.store.shape[2:]) != (H, W):
            # the reference's pool would fail in torch.cat on mixed sizes; here the history buffer has one geometry
            raise GccError('ImagePool holds %dx%d images, queried with %dx%d' % (self.store.shape[2], self.store.shape[3], H, W))
        for gi, lo in enumerate(range(0, N, 8)):
            hi = min(lo + 8, N)
            while len(self._groups) <= gi:
                self._groups.append(_PoolGroup(self))
            grp = self._groups[gi]
            grp.n = hi - lo
            ops.note_dynamic(grp)
            ops.write_i32(self.sel, self._draw(hi - lo))
            ops.image_pool_query(images[lo:hi], out[lo:hi], self.store, self.sel)
        return out


class _PoolGroup:
    """one group of <= 8 images of an ImagePool.query: the unit a recorded gcc_write_i32 launch is patched by"""

    def __init__(self, pool):
        self.pool, self.n = pool, 0

    def replay_update(self, rec, tag):
        """the draws of one more iteration into the recorded gcc_write_i32 launch (its by-value argument 1)"""
        import ctypes as C
        vals = (C.c_int * 16)(*(self.pool._draw(self.n) + [0] * (16 - 2 * self.n)))
        n = ops.lib().gcc_replay_patch(rec, tag, 1, vals, 64)
        if n != 1:
            raise RuntimeError('gcc_replay_patch(image pool tag %d): %d launches patched' % (tag, n))


class _Half:
    """one image set of a batched generator pass: views into the 2N context"""

    def __init__(self, full, lo, hi):
        self.full, self.lo, self.hi = full, lo, hi
        self.N, self.H, self.W = hi - lo, full.H, full.W
        self.out, self.g_out = full.out[lo:hi], full.g_out[lo:hi]


class MobileCycleGANModel(TeacherStreamMixin, nn.Module):

    def __init__(self, opt, cfg_AtoB=None, cfg_BtoA=None):
        super().__init__()
        self.opt = opt
        if len(opt.gpu_ids) == 0 or not torch.cuda.is_available():
            raise GccError('gcc_amd runs on MI355X only (no CPU path): need a visible GPU and gpu_ids >= 0')
        self.device = gdist.local_device(opt)
        ops.lib()
        self.cfg_AtoB, self.cfg_BtoA = cfg_AtoB, cfg_BtoA
        self.loss_names = ['D_A', 'G_A', 'cycle_A', 'idt_A', 'D_B', 'G_B', 'cycle_B', 'idt_B']
        self.visual_names = ['real_A', 'fake_B', 'rec_A', 'idt_B', 'real_B', 'fake_A', 'rec_B', 'idt_A']
        self.generator_extract_layers = ['model.9', 'model.12', 'model.15', 'model.18']
        self.discriminator_extract_layers = ['model.4', 'model.12'] if opt.darts_discriminator else ['model.3', 'model.9']
        self.heavy_sparsity = list(HEAVY_SPARSITY)
        self.teacher_model = None
        dev = self.device

        self.netG_A = MobileResnetGenerator(ngf=opt.ngf, cfg=cfg_AtoB)
        self.netG_B = MobileResnetGenerator(ngf=opt.ngf, cfg=cfg_BtoA)
        self.distill = bool(opt.online_distillation or opt.normal_distillation)
        self.transform_A_convs, self.transform_B_convs = [], []
        if self.distill:
            for cfg, lst in ((cfg_AtoB, self.transform_A_convs), (cfg_BtoA, self.transform_B_convs)):
                s = opt.ngf * 4 if cfg is None else cfg[2]
                lst += [nn.Conv2d(s, opt.teacher_ngf * 4, 1, 1, 0, bias=False).to(dev) for _ in range(4)]
                for t in lst:
                    gdist.broadcast_module(t)      # default-initialised from each rank's RNG: replicas must start equal
        masked = bool(opt.darts_discriminator)
        if masked:
            self.loss_names += ['D_arch_diff_A', 'D_arch_A', 'D_arch_diff_B', 'D_arch_B', 'teacher_netD_A_arch_diff',
                                'teacher_netD_B_arch_diff']
            self.netD_A = MaskNLayerDiscriminator(ndf=opt.ndf, threshold=opt.threshold)
            self.netD_B = MaskNLayerDiscriminator(ndf=opt.ndf, threshold=opt.threshold)
        else:
            self.netD_A = NLayerDiscriminator(ndf=opt.ndf)
            self.netD_B = NLayerDiscriminator(ndf=opt.ndf)
        self.init_net()

        # ---- optimizers over flat parameter groups (:245-343)
        g_params, g_l1 = [], []
        for net, tconvs in ((self.netG_A, self.transform_A_convs), (self.netG_B, self.transform_B_convs)):
            for name, m in net.named_modules():
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                    mult = 1.0 if name not in HEAVY_SPARSITY else (1000.0 if name == 'model.19' else 2.0)
                    for p in m.parameters():
                        g_params.append(p)
                        g_l1.append(opt.lambda_weight * mult if (p.dim() == 4 and opt.lambda_weight > 0.0) else 0.0)
            for t in tconvs:
                g_params.append(t.weight)
                g_l1.append(0.0)
        self.optimizer_G = HipAdam(g_params, lr=opt.lr, betas=(0.5, 0.999), l1=g_l1)
        w_params, a_params = [], []
        for net in (self.netD_A, self.netD_B):
            for m in net.modules():
                if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                    w_params += list(m.parameters())
                elif isinstance(m, DifferentiableOP):
                    a_params += list(m.parameters())
        self.optimizer_D = HipAdam(w_params, lr=opt.lr, betas=(0.5, 0.999))
        if masked:
            self.optimizer_arch = HipAdam(a_params, lr=opt.arch_lr)
            if opt.arch_lr_step:
                arch_opt = copy.deepcopy(opt)
                arch_opt.lr_policy = 'step'
                arch_opt.lr_decay_iters = opt.n_epochs - 1
                self.arch_scheduler = util.get_scheduler(self.optimizer_arch, arch_opt)

        # ---- engines
        self.G = {'A': engine.MobileResnetEngine(self.netG_A, dev), 'B': engine.MobileResnetEngine(self.netG_B, dev)}
        self.D = {'A': engine.PatchGANEngine(self.netD_A, masked, opt.threshold, dev),
                  'B': engine.PatchGANEngine(self.netD_B, masked, opt.threshold, dev)}
        self.T = {'A': [engine.ConvOp(t.weight, None, 1, 1, 0, False) for t in self.transform_A_convs],
                  'B': [engine.ConvOp(t.weight, None, 1, 1, 0, False) for t in self.transform_B_convs]}
        self.refresh_weights()
        self.pool = {'A': ImagePool(50), 'B': ImagePool(50)}     # fake_B_pool feeds D_A, fake_A_pool feeds D_B

        self.optimizers = [self.optimizer_G, self.optimizer_D]
        self.schedulers = [util.get_scheduler(o, opt) for o in self.optimizers]
        if masked and opt.arch_lr_step:
            self.schedulers.append(self.arch_scheduler)
        names = []
        for w in 'AB':
            names += [n + w for n in ('G_', 'cycle_', 'idt_', 'D_real_', 'D_fake_', 'L1_', 'arch_fake_', 'arch_fake_real_',
                                      'arch_real_', 'D_arch_diff_', 'D_arch_', 'teacher_diff_', 'arch_c_fr_', 'arch_c_f_',
                                      's0_', 's1_', 's2_')]
        self._slot = {n: i for i, n in enumerate(names)}
        self._lossvec = torch.zeros(len(names) + 4, dtype=torch.float32, device=dev)
        self._dist_out = {w: torch.zeros((6, 2), dtype=torch.float32, device=dev) for w in 'AB'}
        self._bufs = {}
        self._ema_started = False
        self._world = gdist.world_size()
        self._ctx = None
        self._nchw = {}
        self._dctx_last = {}

    # ---------------------------------------------------------------------------------------
    def _l(self, name):
        i = self._slot[name]
        return self._lossvec[i:i + 1]

    def refresh_weights(self):
        for w in 'AB':
            self.G[w].repack()
            self.D[w].repack()
            for t in self.T[w]:
                t.repack()

    def init_net(self):
        for net in (self.netG_A, self.netG_B, self.netD_A, self.netD_B):
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

    def _forks(self):
        return bool(CYCLE_FORK and not getattr(self, '_no_fork', False) and self._aux_stream())

    def _two_sides(self, side_a, side_b, shared=None):
        """side_a() on the current stream, side_b() on the auxiliary stream beside it (CYCLE_FORK; the online teacher, which
        already runs on a stream of its own, and a single-stream schedule keep both in line); joined before returning.
        shared: the optimizer whose parameter gradients BOTH sides add to (backward_G: each generator is differentiated by
        both sides) -- side B then accumulates into a zeroed second buffer (engine.FlatParams.redirect) that is added
        afterwards: (0 + A) + (0 + B), the bits of A-then-B, and no two launches ever add to one buffer at the same time"""
        aux = self._aux_stream() if self._forks() else False
        if not aux:
            side_a()
            side_b()
            return
        main = ops.current_stream()
        flat = getattr(shared, 'flat', None) if shared is not None else None
        ops.wait_stream(aux, main)
        side_a()
        with ops.on_stream(aux):
            if flat is not None:
                with flat.redirect() as side:
                    ops.fill(side, 0.0)
                    side_b()
            else:
                side_b()
        ops.wait_stream(main, aux)
        if flat is not None:
            ops.add_f32_(flat.grads, flat.side_grads()[0])

    def _g(self, w, tag, src):
        """one generator pass: G_w on the NHWC image ``src``"""
        N, _, H, W = src.shape
        c = self.G[w]._ctx(N, H, W, tag)
        ops.nhwc_copy(src, 0, c.x_in, 0, 3)
        return self.G[w].forward(c)

    def forward(self):
        """six generator passes (:366-380) as four launches sets: G_A(A) and G_A(B) -- the fake and the identity image of
        one generator -- run as ONE pass over a batch of 2N, likewise G_B(B) and G_B(A).  InstanceNorm statistics are per
        image and every other layer is per pixel, so the arithmetic per image is the same as in two passes; at the
        reference's batch size 1 this removes a third of the generator launches, which is what the step time is made
        of.  The contexts keep every activation for backward_G."""
        N, _, H, W = self._A.shape
        c, full = {}, {}

        def side(w, o, first, second, fake, idt, rec):
            cx = self.G[w]._ctx(2 * N, H, W, 'fake_idt')
            ops.nhwc_copy(first, 0, cx.x_in[:N], 0, 3)
            ops.nhwc_copy(second, 0, cx.x_in[N:], 0, 3)
            full[w] = self.G[w].forward(cx)
            c[fake], c[idt] = _Half(full[w], 0, N), _Half(full[w], N, 2 * N)      # G_w(own domain), G_w(other domain)
            c[rec] = self._g(o, 'rec', c[fake].out)                               # the cycle back through the other generator

        # side A: G_A(A) | G_A(B), rec_A = G_B(G_A(A)); side B: G_B(B) | G_B(A), rec_B = G_A(G_B(B)) -- independent chains
        self._two_sides(lambda: side('A', 'B', self._A, self._B, 'fake_B', 'idt_A', 'rec_A'),
                        lambda: side('B', 'A', self._B, self._A, 'fake_A', 'idt_B', 'rec_B'))
        self._ctx = c
        self._nchw = {}

    def _gfeatures(self, w, ctx):
        """hooked generator features of one image set"""
        if isinstance(ctx, _Half):
            return [f[ctx.lo:ctx.hi] for f in self.G[w].features(ctx.full)]
        return self.G[w].features(ctx)

    def visual_forward(self):
        self._ctx = {'fake_B': self._g('A', 'fake', self._A)}
        self._nchw = {}

    def _image(self, name):
        if name not in self._nchw:
            self._nchw[name] = ops.nhwc_to_nchw(self._ctx[name].out, 3)
        return self._nchw[name]

    fake_A = property(lambda self: self._image('fake_A'))
    fake_B = property(lambda self: self._image('fake_B'))
    rec_A = property(lambda self: self._image('rec_A'))
    rec_B = property(lambda self: self._image('rec_B'))
    idt_A = property(lambda self: self._image('idt_A'))
    idt_B = property(lambda self: self._image('idt_B'))

    # -- helpers ------------------------------------------------------------------------------
    def _d_forward(self, w, tag, img):
        N, _, H, W = img.shape
        ctx = self.D[w].new_ctx(N, H, W, tag)
        ops.nhwc_copy(img, 0, ctx.x_in, 0, 3)
        self.D[w].forward(ctx, train=True)
        return ctx

    def _buf(self, key, N, C, H, W):
        key = (key, N, C, H, W)
        if key not in self._bufs:
            self._bufs[key] = ops.new_act(N, C, H, W, self.device)
        return self._bufs[key]

    def _dws(self, key, N, C, HW):
        key = ('ws', key, N, C, HW)
        need = ops.distill_workspace_bytes(N, C, HW)       # depends on the weight-gradient split plan (tuning options)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < need:
            buf = self._bufs[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return buf

    def _allreduce(self, optimizer):
        gdist.all_reduce_grads(optimizer)

    # -- generators (:480-546) ------------------------------------------------------------------------
    def backward_G(self, ts=None):
        opt, mode, c = self.opt, self.opt.gan_mode, self._ctx
        lam = {'A': opt.lambda_A, 'B': opt.lambda_B}
        real = {'A': self._A, 'B': self._B}
        T = self.teacher_model
        # identity terms: idt_A = G_A(B) against B (weight lambda_B), idt_B = G_B(A) against A (weight lambda_A)
        # (their backward runs with the fake pass of the same generator below: one batch of 2N)
        for w, dom in (('A', 'B'), ('B', 'A')):
            ci = c['idt_' + w]
            ops.l1_loss(ci.out, real[dom], self._l('idt_' + w), weight=lam[dom] * opt.lambda_identity, da=ci.g_out)
        # per side: fake = G_w(real), judged by D_w; the cycle through the other generator returns dL/d(fake)
        def side(w, o, fake, rec):
            cf, cr = c[fake], c[rec]
            # the cycle that starts in G_w ends on G_w's input domain: rec_A = G_B(G_A(A)) against A, weight lambda_A
            ops.l1_loss(cr.out, real[w], self._l('cycle_' + w), weight=lam[w], da=cr.g_out)
            dx = self.G[o].backward(cr, need_dx=True)
            ops.nhwc_copy(dx, 0, cf.g_out, 0, 3)
            # GAN term: criterionGAN(D_w(fake), True) keeps for_discriminator's default True (:492-494)
            cd = self._d_forward(w, 'g_fake', cf.out)
            ops.gan_loss(mode, cd.pred, True, True, self._l('G_' + w), dpred=self.D[w].grad_pred_buffer(cd))
            dxd = self.D[w].backward(cd, wgrad=False, need_dx=True)
            ops.nhwc_add(dxd, 0, cf.g_out, 0, 3)
            g_feat = None
            if self.distill:
                self._join(ts)              # first read of the teacher's features (on this side's stream)
                g_feat = self._distill_side(w, cf, T._ctx[fake])
                if w == 'B' and not forked:
                    self._mark_teacher_free()
            self.G[w].backward(cf.full, g_feat=g_feat)

        forked = self._forks()
        self._two_sides(lambda: side('A', 'B', 'fake_B', 'rec_A'), lambda: side('B', 'A', 'fake_A', 'rec_B'), shared=self.optimizer_G)
        if self.distill and forked:
            self._mark_teacher_free()       # both sides have read the teacher (the sides ran on two streams: joined above)

    def _distill_side(self, w, cf, tcf):
        """distillation terms of one generator (:497-541): four transformed generator features carry gradients; the
        teacher discriminator's two features on the (detached) student fake only enter the loss value"""
        opt, T = self.opt, self.teacher_model
        N = cf.N
        ct = T._d_forward(w, 'on_student', cf.out)
        feats = self._gfeatures(w, cf) + T.D[w].features(ct)
        targets = self.target_distillation_A_features if w == 'A' else self.target_distillation_B_features
        g_feat = []
        for i in range(6):
            f, t = feats[i], targets[i]
            if i < 4:
                tf = self._buf(('tf', w, i), N, self.T[w][i].rows, f.shape[2], f.shape[3])
                self.T[w][i].forward(f, tf)
            else:
                tf = f
            ws = self._dws((w, i), N, tf.shape[1], tf.shape[2] * tf.shape[3])
            ops.distill_fwd(tf, t, self._dist_out[w][i], ws, squared=True)
            if i < 4:
                dtf = self._buf(('dtf', w, i), N, tf.shape[1], tf.shape[2], tf.shape[3])
                ops.distill_bwd(tf, t, opt.lambda_gram, opt.lambda_content, dtf, ws, squared=True)
                self.T[w][i].backward_weight(f, dtf)
                # feature gradient of the whole 2N pass: the identity half stays zero
                gbuf = self._buf(('gf', w, i), 2 * N, f.shape[1], f.shape[2], f.shape[3])
                self.T[w][i].backward_data(dtf, gbuf[:N])
                g_feat.append(gbuf)
        if opt.lambda_L1 > 0.0:
            # criterionL1(fake, Tfake) is added once per feature inside the reference's loop: 6 x lambda_L1
            tmp = self._buf(('l1', w), N, 3, cf.H, cf.W)
            ops.l1_loss(cf.out, tcf.out, self._l('L1_' + w), weight=6.0 * opt.lambda_L1, da=tmp)
            ops.nhwc_add(tmp, 0, cf.g_out, 0, 3)
        ops.SideStream.get(self.device).join()
        return g_feat

    # -- discriminators (:382-405): real first, then the pooled fake --------------------------------------
    def backward_D(self):
        mode = self.opt.gan_mode

        def side(w, real, fake):
            img = self._ctx[fake].out
            pooled = self.pool[w].query(img, self._buf(('pool', w), *img.shape))
            cr = self._d_forward(w, 'd_real', real)
            cf = self._d_forward(w, 'd_fake', pooled)
            gp = self.D[w].grad_pred_buffer(cr)
            ops.gan_loss(mode, cr.pred, True, True, self._l('D_real_' + w), dpred=gp, grad_weight=0.5)
            self.D[w].backward(cr, wgrad=True, need_dx=False)
            ops.gan_loss(mode, cf.pred, False, True, self._l('D_fake_' + w), dpred=gp, grad_weight=0.5)
            self.D[w].backward(cf, wgrad=True, need_dx=False)
            self._dctx_last[w] = cf          # what the reference's D hooks hold after the iteration

        self._two_sides(lambda: side('A', self._B, 'fake_B'), lambda: side('B', self._A, 'fake_A'))

    # -- one iteration (:571-590) -------------------------------------------------------------------------
    def optimize_parameters(self):
        with _ChainWgrad():
            return self._optimize_parameters()

    def _optimize_parameters(self):
        ts = None
        if self.opt.online_distillation:
            T = self.teacher_model

            T._no_fork = CYCLE_FORK < 2   # 1: the online teacher, already on a stream of its own, keeps its sides in line

            def teacher_step():
                T.set_input(self.input)
                T.optimize_parameters()
                # the reference clones; here the teacher's activation buffers are not overwritten before they are consumed
                # (read by the student after _join: the step may be enqueued by the teacher's host thread)
                self.target_distillation_A_features = T.get_distillation_features(AorB='A')
                self.target_distillation_B_features = T.get_distillation_features(AorB='B')
            ts = self._run_teacher(teacher_step)
        self.forward()
        self.optimizer_G.zero_grad()
        self.backward_G(ts)
        self._allreduce(self.optimizer_G)
        self.optimizer_G.step()          # L1_sparsity() (:548-569) is fused into the Adam kernel, per-tensor weights
        for w in 'AB':
            self.G[w].repack()
            for t in self.T[w]:
                t.repack()
        self.optimizer_D.zero_grad()
        self.backward_D()
        self._allreduce(self.optimizer_D)
        self.optimizer_D.step()
        for w in 'AB':
            self.D[w].repack()

    # -- architecture step (:407-459, 592-600) ----------------------------------------------------------------
    def get_D_arch_diff(self, isTeacher=False):
        mode = self.opt.gan_mode
        ctxs = {}
        ema = isTeacher and self._ema_started          # one flag for both sides, as the reference tests side A only

        def side(w, fake, real):
            cf = self._d_forward(w, 'a_fake', self._ctx[fake].out)
            cr = self._d_forward(w, 'a_real', real)
            ops.gan_loss(mode, cf.pred, False, True, self._l('arch_fake_' + w))
            ops.gan_loss(mode, cf.pred, True, False, self._l('arch_fake_real_' + w))
            ops.gan_loss(mode, cr.pred, True, True, self._l('arch_real_' + w))
            out = self._l('teacher_diff_' + w if isTeacher else 'D_arch_diff_' + w)
            if ema:
                b = float(self.opt.ema_beta)
                ops.scalar_op(1, self._l('arch_fake_real_' + w), self._l('arch_fake_' + w), out, c=out, k0=b, k1=1.0 - b)
            else:
                ops.scalar_op(0, self._l('arch_fake_real_' + w), self._l('arch_fake_' + w), out)
            ctxs[w] = (cf, cr)

        self._two_sides(lambda: side('A', 'fake_B', self._B), lambda: side('B', 'fake_A', self._A))
        self._ema_started = True
        return ctxs

    def backward_D_arch(self, ts=None):
        T, mode = self.teacher_model, self.opt.gan_mode
        if not ts:
            T.get_D_arch_diff(isTeacher=True)
        ctxs = self.get_D_arch_diff(isTeacher=False)
        self._join(ts)
        for w in 'AB':
            ops.scalar_op(2, T._l('teacher_diff_' + w), T._l('teacher_diff_' + w), self._l('teacher_diff_' + w), k0=0.0)
        self._mark_teacher_free()
        def side(w):
            cf, cr = ctxs[w]
            ops.arch_coeffs(self._l('arch_fake_real_' + w), self._l('arch_fake_' + w), self._l('arch_real_' + w),
                            self._l('teacher_diff_' + w), self._l('D_arch_' + w), self._l('arch_c_fr_' + w),
                            self._l('arch_c_f_' + w))
            gp = self.D[w].grad_pred_buffer(cf)
            ops.gan_loss(mode, cf.pred, True, False, self._l('s0_' + w), dpred=gp, weight_dev=self._l('arch_c_fr_' + w))
            ops.gan_loss(mode, cf.pred, False, True, self._l('s1_' + w), dpred=gp, weight_dev=self._l('arch_c_f_' + w),
                         dpred_accumulate=True)
            self.D[w].backward(cf, wgrad=False, agrad=True, need_dx=False)
            ops.gan_loss(mode, cr.pred, True, True, self._l('s2_' + w), dpred=gp, grad_weight=0.5)
            self.D[w].backward(cr, wgrad=False, agrad=True, need_dx=False)

        self._two_sides(lambda: side('A'), lambda: side('B'))

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
        for net in (self.netD_A, self.netD_B):
            for m in net.modules():
                if isinstance(m, DifferentiableOP):
                    m.clip_alpha()

    # -- bookkeeping surface ----------------------------------------------------------------------
    def print_sparse_info(self, logger):
        for tag, net in (('netD_A', self.netD_A), ('netD_B', self.netD_B)):
            for name, m in net.named_modules():
                if isinstance(m, DifferentiableOP):
                    mask = m.get_current_mask()
                    logger.info('%s %s sparsity ratio: %.2f' % (tag, name, float((mask == 0.0).sum()) / mask.numel()))
            logger.info('-----------------------------------')

    def adaptive_ema_beta(self, epoch):
        self.opt.ema_beta = 1.0 - epoch / (self.opt.n_epochs + self.opt.n_epochs_decay)

    def update_learning_rate(self, epoch):
        for s in self.schedulers:
            s.step()
        self.adaptive_ema_beta(epoch)
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
        ckpt = {'G_A': _portable(self.netG_A.state_dict()), 'G_B': _portable(self.netG_B.state_dict()),
                'D_A': _portable(self.netD_A.state_dict()), 'D_B': _portable(self.netD_B.state_dict()),
                'epoch': epoch, 'cfg': (self.cfg_AtoB, self.cfg_BtoA), 'fid': fid}
        name = 'model_best_%s.pth' % direction if isbest else 'model_%d.pth' % epoch
        torch.save(ckpt, os.path.join(save_dir, name))

    def load_models(self, load_path, load_discriminator=True):
        ckpt = torch.load(load_path, map_location='cpu')
        self.netG_A.load_state_dict(ckpt['G_A'])
        self.netG_B.load_state_dict(ckpt['G_B'])
        if load_discriminator:
            self.netD_A.load_state_dict(ckpt['D_A'])
            self.netD_B.load_state_dict(ckpt['D_B'])
        self.refresh_weights()
        print('loading the model from %s' % load_path)

    def model_train(self):
        for net in (self.netG_A, self.netG_B, self.netD_A, self.netD_B):
            net.train()

    def model_eval(self):
        for net in (self.netG_A, self.netG_B, self.netD_A, self.netD_B):
            net.eval()

    def get_current_visuals(self):
        ret = OrderedDict()
        for name in self.visual_names:
            ret[name] = getattr(self, name)
        return ret

    @property
    def Tfake_A(self):
        return self.teacher_model.fake_A

    @property
    def Tfake_B(self):
        return self.teacher_model.fake_B

    def get_current_losses(self):
        """host read of the device loss scalars (the only sync of the iteration)"""
        v = self._lossvec.cpu()
        d = {w: self._dist_out[w].cpu() for w in 'AB'}
        s = self._slot
        ret = OrderedDict()
        for name in self.loss_names:
            w = name[-1]
            if name in ('D_A', 'D_B'):
                val = 0.5 * (float(v[s['D_real_' + w]]) + float(v[s['D_fake_' + w]]))
            elif name.startswith('content_'):
                val = self.opt.lambda_content * float(d[w][:, 1].sum())
            elif name.startswith('gram_'):
                val = self.opt.lambda_gram * float(d[w][:, 0].sum())
            elif name.startswith('teacher_netD_'):
                val = float(v[s['teacher_diff_' + name[len('teacher_netD_')]]])
            else:
                val = float(v[s[name]])
            ret[name] = val
        if self._world > 1:
            ret = gdist.mean_dict(ret, self.device)
        return ret

    def init_distillation(self):
        if self.distill:
            if self.opt.lambda_content > 0.0:
                self.loss_names += ['content_A', 'content_B']
            if self.opt.lambda_gram > 0.0:
                self.loss_names += ['gram_A', 'gram_B']
            if self.opt.lambda_L1 > 0.0:
                self.loss_names += ['L1_A', 'L1_B']
            self.visual_names += ['Tfake_A', 'Tfake_B']

    def get_distillation_features(self, AorB='A'):
        """4 generator features of the G(real) pass + the 2 discriminator features of the last D call of the iteration
        (the pooled fake of the D step), as the reference's hooks end up holding them"""
        w = AorB
        return self._gfeatures(w, self._ctx['fake_B' if w == 'A' else 'fake_A']) + self.D[w].features(self._dctx_last[w])

    def get_cfg(self):
        return self.cfg_AtoB, self.cfg_BtoA

    # -- pruning (models/CycleGAN.py:794-900): integer logic on host copies of the weights -------------
    def max_min_conv_norm(self, netG):
        from ..utils import prune_util
        return prune_util.max_min_conv_norm_resnet(netG, 'mean')

    def get_prunenet_cfg(self, netG, threshold):
        from ..utils import prune_util
        return prune_util.resnet_prune_cfg(netG, threshold, 'mean')

    def resnet_prune(self, threshold_AtoB, threshold_BtoA):
        cfg_AtoB = self.get_prunenet_cfg(self.netG_A, threshold_AtoB)
        cfg_BtoA = self.get_prunenet_cfg(self.netG_B, threshold_BtoA)
        return MobileCycleGANModel(self.opt, cfg_AtoB=cfg_AtoB, cfg_BtoA=cfg_BtoA)

    def prune(self, threshold, lottery_path=None):
        return self.resnet_prune(threshold, lottery_path)
