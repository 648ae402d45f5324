"""Execution engine of the GCC Pix2Pix / CycleGAN steps on MI355X.

The nn.Module trees in gcc_amd/models only *name and own* parameters (so state_dicts are
interchangeable with the reference, SURVEY.md section 5 "checkpoint").  Everything that computes is
here: explicit forward / backward schedules of HIP kernels (gcc_amd.ops -> libgcc_hip.so) over
persistent NHWC bf16 activation buffers.  There is no autograd tape and no PyTorch operator on the
path; the reference's in-place aliasing (SURVEY.md hazard H1) is made explicit by materialising the
two activated copies each skip tensor is read through (LeakyReLU'd for the next down conv, ReLU'd
inside the concat buffer of the up path).

Reference anchors: U-Net models/Pix2Pix.py:20-130, PatchGAN :267-348, hooks :363-373,702-727.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH

LRELU = 0.2
# run weight-gradient kernels on a side stream, concurrently with the data-gradient / BN chain
OVERLAP_WGRAD = os.environ.get('GCC_OVERLAP_WGRAD', '1') != '0'


def overlap_wgrad_default(world):
    """Weight gradients on the side stream?  Yes on one GPU (+1.2 %); no under data parallelism unless GCC_OVERLAP_WGRAD=1 asks:
    the communication library's stream is one more busy stream on four hardware queues, and with the gradient buckets issued
    from the chains' own streams a data-parallel rank steps in 16.45 ms instead of 17.04-17.30 (one-rank RCCL rig,
    profiles/r5_dp_one_rank.txt; 15.1-15.4 without a process group)"""
    env = os.environ.get('GCC_OVERLAP_WGRAD')
    if env is not None:
        return env != '0'
    return world <= 1
# U-Net layers as one C call (gcc_conv_bn_act): conv + BatchNorm statistics + finalize + normalise / activation
FUSE_CONV_BN = os.environ.get('GCC_FUSE_CONV_BN', '1') != '0'
# GCC_WGRAD_GROUP (default 1): the regular weight gradients of a generator's backward pass run as grouped launches (ops.WgradGroup:
# one launch + one fold per group) instead of one launch + fold per layer -- the U-Net's up path when its last data gradient is
# enqueued, its down path at the end of the pass.  0: per layer, as before round 6.
WGRAD_GROUP = os.environ.get('GCC_WGRAD_GROUP', '1') != '0'
# the power iterations of a SAGAN pass's spectrally normalised layers as four grouped launches (SNConvOp.iterate_all); 0: four per layer
SN_GROUP = os.environ.get('GCC_SN_GROUP', '1') != '0'
WGRAD_GROUP_MAX_UNITS = int(os.environ.get('GCC_WGRAD_GROUP_MAX_UNITS', '20000'))


# ------------------------------------------------------------------------------------------------
# flat fp32 storage of a parameter group: one buffer for values, one for gradients
# ------------------------------------------------------------------------------------------------
class FlatParams:
    """Re-homes parameters into one flat fp32 buffer (values) + one flat gradient buffer, keeping
    every tensor's logical shape and strides (conv weights stay channels_last).  One fill zeroes all
    gradients, one all-reduce exchanges them, one multi-tensor Adam launch updates them."""

    def __init__(self, params, device, layout=None):
        """layout: optional storage order -- a list of lists of parameters (every parameter exactly once), each inner
        list one *segment*; segments are laid down back to back in the order given, which the data-parallel path chooses
        as the order in which the backward pass completes the gradients, so that a finished run of segments is one
        contiguous slice of `grads` (dist.GradReducer: bucketed all-reduce overlapped with the rest of the backward).
        self.segments = [(begin, end)] element ranges.  The parameter list itself (optimizer order) is unchanged."""
        self.params = [p for p in params]
        if layout is None:
            layout = [self.params]
        placed = [p for seg in layout for p in seg]
        assert len(placed) == len(self.params) and {id(p) for p in placed} == {id(p) for p in self.params}, \
            'layout must hold every parameter exactly once'
        where, total, self.segments = {}, 0, []
        for seg in layout:
            b = total
            for p in seg:
                where[id(p)] = total
                total += (p.numel() + 3) & ~3
            self.segments.append((b, total))
        offs = [where[id(p)] for p in self.params]
        self.total = total
        self.values = torch.zeros(max(total, 4), dtype=torch.float32, device=device)
        self.grads = torch.zeros(max(total, 4), dtype=torch.float32, device=device)
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                src = p.detach().to(device)
                if src.dim() == 4 and src.shape[2] * src.shape[3] > 1:
                    src = src.contiguous(memory_format=torch.channels_last)
                else:
                    src = src.contiguous()
                v = torch.as_strided(self.values, src.shape, src.stride(), o)
                v.copy_(src)
                p.data = v
                g = torch.as_strided(self.grads, src.shape, src.stride(), o)
                p.grad = g
                self.grad_views.append(g)

    def zero_grad(self):
        ops.fill(self.grads, 0.0)

    # -- a second accumulator (two chains that add to the same gradients side by side on two streams) ---------------------------
    def side_grads(self):
        """(buffer shaped like `grads`, views of it shaped / strided like every parameter's .grad)"""
        if getattr(self, '_side', None) is None:
            buf = torch.zeros_like(self.grads)
            base = self.grads.data_ptr()
            views = [torch.as_strided(buf, g.shape, g.stride(), (g.data_ptr() - base) // 4) for g in self.grad_views]
            self._side = (buf, views)
        return self._side

    class _Redirect:
        def __init__(self, flat):
            self.flat = flat

        def __enter__(self):
            for p, v in zip(self.flat.params, self.flat.side_grads()[1]):
                p.grad = v
            return self.flat.side_grads()[0]

        def __exit__(self, *exc):
            for p, g in zip(self.flat.params, self.flat.grad_views):
                p.grad = g
            return False

    def redirect(self):
        """with flat.redirect() as side: ... -- launches enqueued inside accumulate parameter gradients into `side` (which the
        caller zeroes first and adds to `grads` afterwards: gcc_add_f32) instead of `grads`.  Host-side pointer swap: the
        kernels take the address at enqueue time."""
        return FlatParams._Redirect(self)


def bn_buffers_to(module, device):
    for b in module.buffers():
        b.data = b.data.to(device)


# ------------------------------------------------------------------------------------------------
class ConvOp:
    """A Conv2d or ConvTranspose2d parameter set with its bf16 packings.

    master weight [rows, cols, k, k] (channels_last):  Conv2d rows=Co cols=Ci ;
    ConvTranspose2d rows=Cin cols=Cout == the adjoint Conv2d (big image -> small image) with
    Co=rows, Ci=cols; its forward is that conv's backward-data."""

    def __init__(self, weight, bias, k, stride, pad, transposed, row_split=0, col_split=0):
        """row_split / col_split: the master's row / column dimension is a channel concatenation whose
        first part has that many channels (each part padded to 8 in memory); 0 = a single tensor."""
        self.weight, self.bias = weight, bias
        self.k, self.stride, self.pad, self.transposed = k, stride, pad, transposed
        self.rows, self.cols = weight.shape[0], weight.shape[1]
        # a split on a multiple of 8 needs no padding: treat as unsplit (keeps the direct wgrad path)
        self.row_split = row_split if (0 < row_split < self.rows and row_split % 8) else 0
        self.col_split = col_split if (0 < col_split < self.cols and col_split % 8) else 0
        self.rows_p = ops.seg_phys(self.rows, self.row_split)
        self.cols_p = ops.seg_phys(self.cols, self.col_split)
        # channel counts the kernels are told: physical when the dimension is a padded concatenation
        self.rows_k = self.rows_p if self.row_split else self.rows
        self.cols_k = self.cols_p if self.col_split else self.cols
        dev = weight.device
        taps = k * k
        self.w = torch.zeros((self.rows_p, taps, self.cols_p), dtype=torch.bfloat16, device=dev)
        self.wt = torch.zeros((self.cols_p, taps, self.rows_p), dtype=torch.bfloat16, device=dev)
        self._pack = None
        self._groupable = {}

    def repack(self):
        if self._pack is None:
            self._pack = ops.PackPlan([self], self.weight.device)
        self._pack.run()

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x, out, act=ACT_NONE, want_stats=False, use_bias=True, bn=None, y2=None, y2_mode=0, y2_gate=None):
        """bn: gcc_bn_t (BNOp.desc) of the BatchNorm behind this conv -- finalized inside the call (with want_stats);
        y2 / y2_mode / y2_gate: a second output (ops.conv_fprop; plain convs only)"""
        b = self.bias.data if (self.bias is not None and use_bias) else None
        if not self.transposed:
            return ops.conv_fprop(x, self.w, self.rows_k, self.k, self.stride, self.pad, out=out, bias=b, act=act,
                                  slope=LRELU, want_stats=want_stats, bn=bn, y2=y2, y2_mode=y2_mode, y2_gate=y2_gate)
        assert y2 is None
        N, _, H, W = out.shape
        return ops.conv_dgrad(x, self.wt, self.cols_k, H, W, self.k, self.stride, self.pad, out=out, bias=b, act=act,
                              slope=LRELU, want_stats=want_stats, bn=bn)

    def forward_bn_act(self, x, raw, bnop, st, count, y, y2=None, act=ACT_NONE, act2=ACT_NONE, drop_p=0.0, seed=0):
        """conv + BatchNorm (training statistics) + activation [+ dropout, + second activated copy] as one C call"""
        w, dgrad = (self.wt, True) if self.transposed else (self.w, False)
        ops.conv_bn_act(dgrad, x, w, raw, self.k, self.stride, self.pad, bnop.bn, st, count, y, y2, act=act, act2=act2,
                        slope=LRELU, drop_p=drop_p, seed=seed)
        bnop.pending_batches += 1
        ops.note_host(bnop)

    # -- gradient w.r.t. the layer input ------------------------------------------------------
    def backward_data(self, dy, out):
        if not self.transposed:
            N, _, H, W = out.shape
            return ops.conv_dgrad(dy, self.wt, self.cols_k, H, W, self.k, self.stride, self.pad, out=out)
        return ops.conv_fprop(dy, self.w, self.rows_k, self.k, self.stride, self.pad, out=out)

    # -- gradient w.r.t. the weight (accumulates into weight.grad) and bias -------------------------
    def backward_weight(self, x, dy, bias_done=False):
        """bias_done: the bias gradient (channel sums of dy) was already accumulated by the kernel that produced dy"""
        if OVERLAP_WGRAD:
            side = ops.SideStream.get(x.device)
            side.fork()
            with ops.on_stream(side.stream):
                self._backward_weight(x, dy, bias_done)
        else:
            self._backward_weight(x, dy, bias_done)

    def group_entry(self, x, dy):
        """this layer's weight gradient as an entry of ops.WgradGroup, or None when it has to run on its own (a padded channel
        concatenation, an irregular width); a bias gradient stays a launch of its own (WgradCollector)"""
        if self.row_split or self.col_split or (self.cols & 7) or self.weight.grad is None:
            return None
        cx, cdy = (x, dy) if not self.transposed else (dy, x)
        e = (cx, cdy, self.weight.grad, self.k, self.stride, self.pad, True)
        # only layers whose own launch cannot fill the chip: (128 x 128 output tiles) x (64-pixel steps) of work.  A larger layer's
        # launch is efficient by itself and overlaps the chain; grouped it waits for the flush (SRGAN's trunk at 96 x 96 x 16
        # images, 11.5 k units per layer: +1.1 % on the 96 -> 384 step when grouped, profiles/r6_wgrad_group.txt)
        units = -(-cdy.shape[0] * cdy.shape[2] * cdy.shape[3] // 64) * -(-self.k * self.k * cx.shape[1] // 128) * -(-cdy.shape[1] // 128)
        if units > WGRAD_GROUP_MAX_UNITS:
            return None
        # the library's own answer for this geometry (a thin-output or head layer has a route of its own), asked once per shape
        ok = self._groupable.get((tuple(cx.shape), tuple(cdy.shape), self.weight.grad.data_ptr() & 15))
        if ok is None:
            ok = self._groupable[(tuple(cx.shape), tuple(cdy.shape), self.weight.grad.data_ptr() & 15)] = bool(ops.wgrad_groupable(*e[:6]))
        return e if ok else None

    def _backward_weight(self, x, dy, bias_done=False):
        cx, cdy = (x, dy) if not self.transposed else (dy, x)      # (conv input, conv output-gradient) of the adjoint pair
        if self.row_split or self.col_split:
            ops.conv_wgrad_seg(cx, cdy, self.weight.grad, self.rows, self.cols, self.row_split, self.col_split, self.k,
                               self.stride, self.pad, accumulate=True)
        else:
            ops.conv_wgrad(cx, cdy, self.weight.grad, self.k, self.stride, self.pad, accumulate=True)
        if self.bias is not None and not bias_done:
            ops.channel_sum(dy, self.bias.grad, accumulate=True)


class BNOp:
    def __init__(self, bn: nn.BatchNorm2d):
        self.bn = bn
        self.C = bn.num_features
        self.pending_batches = 0           # num_batches_tracked increments not yet written to the buffer
        bn.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.flush_counter())

    def flush_counter(self):
        """num_batches_tracked is bookkeeping only (momentum is fixed): it is counted on the host and
        written into the module's buffer when a state_dict is taken, instead of one tiny kernel per
        BatchNorm application."""
        if self.pending_batches:
            self.bn.num_batches_tracked += self.pending_batches
            self.pending_batches = 0

    def replay_update(self, rec, tag):
        self.pending_batches += 1          # one more training-mode application inside a replayed iteration

    def desc(self, st, count, device, running=True):
        """gcc_bn_t for a conv call that finalizes this (training-mode) BatchNorm itself -- ConvOp.forward(..., bn=...): no
        gcc_bn_finalize launch on the chain.  Counts the application like finalize() does."""
        self.pending_batches += 1
        ops.note_host(self)
        return ops.bn_desc(self.bn, st, count, device, running)

    def finalize(self, stats, count, st, train):
        bn = self.bn
        if train:
            ops.bn_finalize(stats, count, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, st,
                            eps=bn.eps, momentum=bn.momentum)
            self.pending_batches += 1
            ops.note_host(self)
        else:
            ops.bn_eval_coeffs(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, st, eps=bn.eps)


def _get(module, dotted):
    m = module
    for part in dotted.split('.'):
        m = getattr(m, part)
    return m


def refresh_gate_masks(obj):
    """mask_i = (sign(alpha_i - tau) + 1) / 2 for every gated layer of `obj` (obj.gate / obj.mask / obj.tau).  Once the arch
    optimizer has re-homed the alphas into its flat buffer they are neighbours in one storage: one launch over the whole
    span (the masks become views of one buffer at the same offsets) instead of one launch per layer -- these launches sit
    on the student's critical chain, 4 per discriminator pass.  The alphas' addresses are re-checked on every call."""
    gates = [(i, g) for i, g in enumerate(obj.gate) if g is not None]
    if not gates:
        return
    alphas = [g.alpha.data for _, g in gates]
    if getattr(obj, 'mask_cache', False):
        # the owner promises to set mask_dirty whenever it changes an alpha through the library (arch optimizer step,
        # clip_alpha, weight reload); writes made with torch operators show up in the tensors' version counters
        stamp = tuple(g.alpha._version for _, g in gates) + tuple(a.data_ptr() for a in alphas)
        if not getattr(obj, 'mask_dirty', True) and getattr(obj, '_mask_stamp', None) == stamp:
            return
        obj._mask_stamp, obj.mask_dirty = stamp, False
    key = tuple(a.data_ptr() for a in alphas)
    if getattr(obj, '_mask_key', None) != key:
        obj._mask_key = key
        obj._mask_span = None
        a0 = alphas[0]
        base = a0.untyped_storage().data_ptr()
        same = len(alphas) > 1 and all(a.dtype == torch.float32 and a.is_contiguous() and
                                       a.untyped_storage().data_ptr() == base for a in alphas)
        lo = min(a.storage_offset() for a in alphas)
        hi = max(a.storage_offset() + a.numel() for a in alphas)
        if same and lo % 4 == 0 and hi - lo <= 2 * sum(a.numel() for a in alphas) + 64:
            obj._alpha_span = torch.empty(0, dtype=torch.float32, device=a0.device).set_(a0.untyped_storage(), lo, (hi - lo,), (1,))
            obj._mask_span = torch.ones(hi - lo, dtype=torch.float32, device=a0.device)
            for (i, _), a in zip(gates, alphas):
                o = a.storage_offset() - lo
                obj.mask[i] = obj._mask_span[o:o + a.numel()]
        else:
            for (i, _), a in zip(gates, alphas):
                obj.mask[i] = torch.ones(a.numel(), dtype=torch.float32, device=a.device)
    if obj._mask_span is not None:
        ops.gate_mask(obj._alpha_span, obj.tau, obj._mask_span)
    else:
        for (i, _), a in zip(gates, alphas):
            ops.gate_mask(a, obj.tau, obj.mask[i])


class WgradCollector:
    """The weight gradients of one backward pass, collected and run as grouped launches (ops.WgradGroup: one launch + one fold for
    all regular layers added since the last flush) on the stream the per-layer launches would have taken (the weight-gradient side
    stream when engine.OVERLAP_WGRAD).  A layer the group cannot take (ConvOp.group_entry) runs at once, as before; bias gradients
    stay channel sums of their own, issued with the group.  The tables live on the pass's context object (one per context and
    flush point: the operands' addresses are part of a table).  owner._seg_done(seg), where given, reports the segment to the
    data-parallel reducer behind the launch that completes it."""

    def __init__(self, owner, ctx, enabled=True):
        self.owner, self.ctx = owner, ctx
        self.on = bool(enabled) and WGRAD_GROUP and not ops.PROFILE.active      # (bench.py's bracketed step times every launch on its own)
        self.entries, self.bias, self.segs = [], [], []

    def add(self, conv, x, dy, seg=None, bias_done=False):
        e = conv.group_entry(x, dy) if self.on else None
        if e is None:
            conv.backward_weight(x, dy, bias_done) if bias_done else conv.backward_weight(x, dy)
            if seg is not None:
                # segments are reported in completion order (dist.GradReducer sends a bucket when its LAST segment is reported):
                # behind collected layers that have not been launched yet, this one's report waits for their flush
                if self.entries:
                    self.segs.append(seg)
                else:
                    self.owner._seg_done(seg)
            return
        self.entries.append(e)
        if conv.bias is not None and not bias_done:
            self.bias.append((dy, conv.bias.grad))
        if seg is not None:
            self.segs.append(seg)

    def flush(self, name):
        entries, bias, segs = self.entries, self.bias, self.segs
        self.entries, self.bias, self.segs = [], [], []
        if not entries:
            return
        groups = self.ctx.__dict__.setdefault('wgrad_groups', {})

        def launch():
            for i in range(0, len(entries), _lib.WGRAD_GROUP_MAX):
                part = entries[i:i + _lib.WGRAD_GROUP_MAX]
                key = (name, i)
                g = groups.get(key)
                if g is None:
                    g = groups[key] = ops.WgradGroup()
                if len(part) > 1 and g.groupable(part):
                    g.run()
                else:
                    for (x, dy, dw, k, stride, pad, acc) in part:
                        ops.conv_wgrad(x, dy, dw, k, stride, pad, accumulate=acc)
            if len(bias) > 1:
                ops.channel_sum_group(bias, accumulate=True)
            else:
                for dy, bg in bias:
                    ops.channel_sum(dy, bg, accumulate=True)
        if OVERLAP_WGRAD:
            side = ops.SideStream.get(self.owner.device)
            side.fork()
            with ops.on_stream(side.stream):
                launch()
        else:
            launch()
        for seg in segs:
            self.owner._seg_done(seg)


# ------------------------------------------------------------------------------------------------
# U-Net generator
# ------------------------------------------------------------------------------------------------
class UnetEngine:
    """Forward / backward schedule of UnetGenertor(num_downs=D) for channel widths that are
    multiples of 8 (pruned, irregular widths: see DESIGN.md 'next')."""

    def __init__(self, module, num_downs, device, use_dropout=False):
        # D = number of blocks that are BUILT (a pruned generator may have lost inner blocks: UnetGenertor.present); everything
        # below goes by nesting position, as the reference's module names do
        D = len(getattr(module, 'present', range(num_downs)))
        self.module, self.D, self.device = module, D, device
        self.use_dropout = use_dropout
        # last block = a loop block around Identity (models/Pix2Pix.py:59-67) instead of the innermost kind: conv, BatchNorm, ReLU
        self.inner_identity = bool(getattr(module, 'inner_identity', False))
        ii = self.inner_identity

        def prefix(d):
            return 'model' if d == 0 else 'model.model.1' + '.model.3' * (d - 1)
        self.prefix = prefix
        self.down, self.down_bn, self.up, self.up_bn = [None] * D, [None] * D, [None] * D, [None] * D
        for d in range(D):
            p = prefix(d)
            cname = p + ('.model.0' if d == 0 else '.model.1')
            c = _get(module, cname)
            self.down[d] = ConvOp(c.weight, c.bias, 4, 2, 1, False)
            if 0 < d < D - 1 or (d == D - 1 and ii):
                self.down_bn[d] = BNOp(_get(module, p + '.model.2'))
            uname = p + ('.model.3' if (d == 0 or (d == D - 1 and not ii)) else '.model.5')
            u = _get(module, uname)
            # the up conv at depth d < D-1 reads cat(skip e[d] | up output of depth d+1): rows are a concatenation
            self.up[d] = ConvOp(u.weight, u.bias, 4, 2, 1, True, row_split=(self.down[d].rows if d < D - 1 else 0))
            if d > 0:
                self.up_bn[d] = BNOp(_get(module, p + ('.model.4' if (d == D - 1 and not ii) else '.model.6')))
        self.width = [self.down[d].rows for d in range(D)]          # channels of e[d]
        self.uwidth = [self.up[d].cols for d in range(D)]           # channels of the up-conv output at depth d
        # channel offset of the up-path part inside the concat buffer of depth d (skip part padded to 8)
        self.uoff = [0] + [ops.ceil8(self.width[d - 1]) for d in range(1, D)]
        self.catw = [0] + [self.uoff[d] + ops.ceil8(self.uwidth[d]) for d in range(1, D)]    # physical concat width
        # channel count the kernels are told for the concat buffer: the physical width when the skip part is
        # padded inside it, else the plain sum (same convention as ConvOp.rows_k / cols_k)
        self.catc = [0] + [self.catw[d] if self.width[d - 1] % 8 else self.width[d - 1] + self.uwidth[d]
                           for d in range(1, D)]
        self.drop_depths = ([D - 2 - i for i in range(D - 5)] if not hasattr(module, 'dropout_positions')
                            else list(module.dropout_positions)) if use_dropout else []
        if D < 5 and not (D == 4 and ii):
            raise NotImplementedError('U-Net of %d blocks' % D)
        self.hook_names = ['model.model.1.model.2', 'model.model.1.model.3.model.3.model.2',
                           'model.model.1.model.3.model.3.model.4', 'model.model.1.model.4']
        self.ctx = {}
        self.seed = 0x5EED

    def convs(self):
        return [c for c in self.down + self.up if c is not None]

    @staticmethod
    def grad_segments(module, num_downs):
        """parameters grouped in the order _backward() completes their gradients (one segment per layer: conv weight / bias
        with the BatchNorm behind it): up[0], up[1] .. up[D-1], down[D-1] .. down[0].  Same name scheme as __init__."""
        D = len(getattr(module, 'present', range(num_downs)))
        ii = bool(getattr(module, 'inner_identity', False))
        prefix = lambda d: 'model' if d == 0 else 'model.model.1' + '.model.3' * (d - 1)
        par = lambda name: [p for p in _get(module, name).parameters()]
        segs = []
        for d in range(D):
            p = prefix(d)
            inner = d == D - 1 and not ii
            seg = par(p + ('.model.3' if (d == 0 or inner) else '.model.5'))
            if d > 0:
                seg = par(p + ('.model.4' if inner else '.model.6')) + seg
            segs.append(seg)
        for d in range(D - 1, -1, -1):
            p = prefix(d)
            seg = par(p + ('.model.0' if d == 0 else '.model.1'))
            if 0 < d < D - 1 or (d == D - 1 and ii):
                seg = par(p + '.model.2') + seg
            segs.append(seg)
        return segs

    def _seg_done(self, i):
        """segment i of grad_segments() is complete once everything enqueued so far (main stream + weight-gradient side
        stream) has run: report it to the data-parallel reducer, ordered behind the side stream"""
        r = getattr(self, 'reducer', None)
        if r is not None:
            r.segment_done(self.seg_base + i, ops.SideStream.get(self.device).stream if OVERLAP_WGRAD else None)

    seg_base = 0          # index of this engine's first segment in the optimizer's layout (transform convs come first)

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.convs(), self.device)
        self._pack.run()

    # ---------------------------------------------------------------------------------------
    def _ctx(self, N, H, W, slot=0):
        key = (N, H, W) if slot == 0 else (N, H, W, slot)       # slot: a second set of activations (a forward that must not
        if key in self.ctx:                                       # overwrite features another stream still reads)
            return self.ctx[key]
        D, dev, wd, uw = self.D, self.device, self.width, self.uwidth
        c = type('UnetCtx', (), {})()
        c.N, c.H, c.W = N, H, W
        hs = [(H >> (d + 1), W >> (d + 1)) for d in range(D)]     # spatial size of e[d]
        c.hs = hs
        c.x_in = ops.new_act(N, 3, H, W, dev)
        c.e = [None] * D        # raw conv output (pre-BN) at depth d (d = D-1: post-ReLU, fused)
        c.lin = [None] * (D + 1)  # lin[d] = leaky_relu(e[d-1]) : input of down conv d
        c.rcat = [None] * D     # rcat[d] = relu(cat(e[d-1] | u_d)) : input of up conv d-1
        c.t = [None] * D        # raw up-conv output (pre-BN) of depth d, d >= 1
        c.st_down = [None] * D
        c.st_up = [None] * D
        for d in range(D):
            h, w = hs[d]
            c.e[d] = ops.new_act(N, wd[d], h, w, dev)
            if d < D - 1:
                c.lin[d + 1] = ops.new_act(N, wd[d], h, w, dev)
                c.rcat[d + 1] = ops.new_act(N, self.catc[d + 1], h, w, dev, ld=self.catw[d + 1])
            if 0 < d < D - 1 or (d == D - 1 and self.inner_identity):
                c.st_down[d] = ops.BNState(wd[d], dev)
            if d >= 1:
                hh, ww = hs[d - 1]
                c.t[d] = ops.new_act(N, uw[d], hh, ww, dev)
                c.st_up[d] = ops.BNState(uw[d], dev)
        # last block around Identity: relu(bn(e[D-1])) is what its transposed conv reads (the innermost kind fuses the ReLU
        # into the conv and reads e[D-1] itself)
        c.e_act = ops.new_act(N, wd[D - 1], hs[D - 1][0], hs[D - 1][1], dev) if self.inner_identity else c.e[D - 1]
        c.out = ops.new_act(N, 3, H, W, dev)
        # gradient buffers
        c.g_out = ops.new_act(N, 3, H, W, dev)
        c.g_rcat = [None] * D
        c.g_lin = [None] * (D + 1)
        c.g_t = [None] * D
        for d in range(D):
            h, w = hs[d]
            if d < D - 1:
                c.g_rcat[d + 1] = ops.new_act(N, self.catc[d + 1], h, w, dev, ld=self.catw[d + 1])
                c.g_lin[d + 1] = ops.new_act(N, wd[d], h, w, dev)
            if d >= 1:
                hh, ww = hs[d - 1]
                c.g_t[d] = ops.new_act(N, uw[d], hh, ww, dev)
        c.g_e_last = ops.new_act(N, wd[D - 1], hs[D - 1][0], hs[D - 1][1], dev)
        c.g_e_in = ops.new_act(N, wd[D - 1], hs[D - 1][0], hs[D - 1][1], dev) if self.inner_identity else None
        c.g_e = [ops.new_act(N, wd[d], hs[d][0], hs[d][1], dev) for d in range(D - 1)]
        c.train = True
        c.iter_seed = 0
        self.ctx[key] = c
        return c

    def features(self, c):
        """the four hooked tensors, in the reference's order.  Four blocks only (everything below depth 3 pruned away): the
        BatchNorm output of block 3 is overwritten in place by that block's own ReLU (Identity hands the same tensor on), so
        both of its hooks see relu(bn(e[3])) (hazard H1)"""
        if self.D == 4:
            return [c.lin[2], c.e_act, c.e_act, c.rcat[2]]
        return [c.lin[2], c.lin[4], c.rcat[4], c.rcat[2]]

    # ---------------------------------------------------------------------------------------
    def forward(self, N, H, W, train=True, slot=0):
        """x must already sit in ctx.x_in (use _ctx(N,H,W,slot).x_in); returns ctx (ctx.out = tanh image)."""
        tag = getattr(self, 'profile_tag', None)
        if tag and (ops.PROFILE.active or ops.PROFILE.spans_only):             # bench.py's roofline.generator block
            with ops.PROFILE.span(tag + '.fwd'):
                return self._forward(N, H, W, train, slot)
        return self._forward(N, H, W, train, slot)

    # timing ablation only (scratch/ablate_generators.py): True = forward / backward enqueue nothing and the buffers keep the
    # values of the last real pass -- bounds what the U-Net passes cost the production schedule (results are stale by design)
    ablate_skip = False

    def _forward(self, N, H, W, train=True, slot=0):
        if self.ablate_skip and ((N, H, W) if slot == 0 else (N, H, W, slot)) in self.ctx:
            return self._ctx(N, H, W, slot)
        c = self._ctx(N, H, W, slot)
        D, wd, uw = self.D, self.width, self.uwidth
        c.train = train
        self.seed += 1
        c.iter_seed = self.seed
        # ---- down path
        # outermost down conv (no norm): both activated copies of e[0] straight from the conv launch -- the LeakyReLU'd one for the
        # next down conv, the ReLU'd one into the concat buffer (hazard H1); e[0] itself is never materialised
        self.down[0].forward(c.x_in, c.lin[1], act=ACT_LRELU, y2=ops.cslice(c.rcat[1], 0, wd[0]), y2_mode=ops.Y2_RELU)
        # (bench.py's bracketed roofline step times every igemm launch on its own: the layers run as separate calls there)
        fused = train and FUSE_CONV_BN and not ops.PROFILE.active
        for d in range(1, D - 1):
            n = N * c.hs[d][0] * c.hs[d][1]
            if fused and self.down[d].bias is None and not (self.down[d].row_split or self.down[d].col_split):
                self.down[d].forward_bn_act(c.lin[d], c.e[d], self.down_bn[d], c.st_down[d], n, c.lin[d + 1],
                                            ops.cslice(c.rcat[d + 1], 0, wd[d]), act=ACT_LRELU, act2=ACT_RELU)
                continue
            if train:
                self.down[d].forward(c.lin[d], c.e[d], want_stats=True, bn=self.down_bn[d].desc(c.st_down[d], n, self.device))
            else:
                _, stats = self.down[d].forward(c.lin[d], c.e[d], want_stats=True)
                self.down_bn[d].finalize(stats, n, c.st_down[d], train)
            ops.bnact_fwd(c.e[d], c.lin[d + 1], ops.cslice(c.rcat[d + 1], 0, wd[d]), scale=c.st_down[d].scale,
                          shift=c.st_down[d].shift, act=ACT_LRELU, act2=ACT_RELU)
        if not self.inner_identity:
            self.down[D - 1].forward(c.lin[D - 1], c.e[D - 1], act=ACT_RELU)      # innermost: conv + ReLU fused
        else:                    # a loop block around Identity: conv, BatchNorm, (in-place) ReLU
            d, n = D - 1, N * c.hs[D - 1][0] * c.hs[D - 1][1]
            if fused and self.down[d].bias is None and not (self.down[d].row_split or self.down[d].col_split):
                self.down[d].forward_bn_act(c.lin[d], c.e[d], self.down_bn[d], c.st_down[d], n, c.e_act, None, act=ACT_RELU)
            else:
                if train:
                    self.down[d].forward(c.lin[d], c.e[d], want_stats=True, bn=self.down_bn[d].desc(c.st_down[d], n, self.device))
                else:
                    _, stats = self.down[d].forward(c.lin[d], c.e[d], want_stats=True)
                    self.down_bn[d].finalize(stats, n, c.st_down[d], train)
                ops.bnact_fwd(c.e[d], c.e_act, scale=c.st_down[d].scale, shift=c.st_down[d].shift, act=ACT_RELU)
        # ---- up path
        src = c.e_act
        for d in range(D - 1, 0, -1):
            hh, ww = c.hs[d - 1]
            drop = 0.5 if (train and d in self.drop_depths) else 0.0
            if fused and self.up[d].bias is None and not (self.up[d].row_split or self.up[d].col_split):
                self.up[d].forward_bn_act(src, c.t[d], self.up_bn[d], c.st_up[d], N * hh * ww,
                                          ops.cslice(c.rcat[d], self.uoff[d], uw[d]), act=ACT_RELU, drop_p=drop,
                                          seed=c.iter_seed * 64 + d)
                src = c.rcat[d]
                continue
            if train:
                self.up[d].forward(src, c.t[d], want_stats=True, bn=self.up_bn[d].desc(c.st_up[d], N * hh * ww, self.device))
            else:
                _, stats = self.up[d].forward(src, c.t[d], want_stats=True)
                self.up_bn[d].finalize(stats, N * hh * ww, c.st_up[d], train)
            ops.bnact_fwd(c.t[d], ops.cslice(c.rcat[d], self.uoff[d], uw[d]), scale=c.st_up[d].scale,
                          shift=c.st_up[d].shift, act=ACT_RELU, drop_p=drop, seed=c.iter_seed * 64 + d)
            src = c.rcat[d]
        self.up[0].forward(c.rcat[1], c.out, act=ACT_TANH)
        return c

    # ---------------------------------------------------------------------------------------
    def backward(self, c, g_feat=None, wgrad=True):
        """c.g_out holds dL/d(out).  g_feat: optional list of 4 gradients w.r.t. features(c).
        Accumulates parameter gradients (wgrad) ; nothing is returned (the input image needs none)."""
        tag = getattr(self, 'profile_tag', None)
        if tag and (ops.PROFILE.active or ops.PROFILE.spans_only):
            with ops.PROFILE.span(tag + '.bwd'):
                return self._backward(c, g_feat, wgrad)
        return self._backward(c, g_feat, wgrad)

    def _backward(self, c, g_feat=None, wgrad=True):
        if self.ablate_skip:
            return
        D, wd, uw, N = self.D, self.width, self.uwidth, c.N
        if g_feat is None:
            g_feat = [None] * 4
        # weight gradients: the regular layers are collected and run as two grouped launches (WGRAD_GROUP; bench.py's bracketed
        # roofline step times every launch on its own: per layer there)
        wg = WgradCollector(self, c)
        weight_gradient = wg.add
        # outermost: tanh' then the transposed conv
        ops.bnact_bwd(c.out, None, c.g_out, c.g_out, in_act=ACT_TANH)
        if wgrad:
            weight_gradient(self.up[0], c.rcat[1], c.g_out, 0)
        self.up[0].backward_data(c.g_out, c.g_rcat[1])
        for d in range(1, D):
            if d == 2 and g_feat[3] is not None:
                ops.nhwc_add(g_feat[3], 0, c.g_rcat[2], 0, self.catc[2])
            if d == 4 and g_feat[2] is not None:
                ops.nhwc_add(g_feat[2], 0, c.g_rcat[4], 0, self.catc[4])
            w = wd[d - 1]
            bn = self.up_bn[d].bn
            drop = 0.5 if (c.train and d in self.drop_depths) else 0.0
            # y = None: the activation's sign is recomputed from the BatchNorm affine instead of reading the saved output
            ops.bnact_bwd(c.t[d], None, ops.cslice(c.g_rcat[d], self.uoff[d], uw[d]), c.g_t[d], bn=c.st_up[d],
                          gamma=bn.weight.data, beta=bn.bias.data, bn_eval=not c.train, act=ACT_RELU, drop_p=drop,
                          seed=c.iter_seed * 64 + d, dgamma=bn.weight.grad if wgrad else None,
                          dbeta=bn.bias.grad if wgrad else None)
            src = c.e_act if d == D - 1 else c.rcat[d + 1]
            if wgrad:
                weight_gradient(self.up[d], src, c.g_t[d], d)
            self.up[d].backward_data(c.g_t[d], c.g_e_last if d == D - 1 else c.g_rcat[d + 1])
        # the up path's gradients are all there: its group runs beside the down path's chain
        wg.flush('up')
        if not self.inner_identity:
            # innermost down conv (+ fused ReLU)
            ops.bnact_bwd(c.e[D - 1], None, c.g_e_last, c.g_e_last, in_act=ACT_RELU)
        else:
            # loop block around Identity: ReLU + BatchNorm backward (four blocks: block 3's two hooked tensors are this ReLU's output)
            if D == 4:
                for gf in (g_feat[1], g_feat[2]):
                    if gf is not None:
                        ops.nhwc_add(gf, 0, c.g_e_last, 0, wd[D - 1])
            bn = self.down_bn[D - 1].bn
            ops.bnact_bwd(c.e[D - 1], None, c.g_e_last, c.g_e_in, bn=c.st_down[D - 1], gamma=bn.weight.data, beta=bn.bias.data,
                          bn_eval=not c.train, act=ACT_RELU, dgamma=bn.weight.grad if wgrad else None,
                          dbeta=bn.bias.grad if wgrad else None)
        g_in = c.g_e_in if self.inner_identity else c.g_e_last
        if wgrad:
            weight_gradient(self.down[D - 1], c.lin[D - 1], g_in, D)
        self.down[D - 1].backward_data(g_in, c.g_lin[D - 1])
        for d in range(D - 2, -1, -1):
            # e[d] feeds lin[d+1] (LeakyReLU) and rcat[d+1][:w] (ReLU)
            if d == 1 and g_feat[0] is not None:
                ops.nhwc_add(g_feat[0], 0, c.g_lin[2], 0, wd[1])
            if d == 3 and g_feat[1] is not None:
                ops.nhwc_add(g_feat[1], 0, c.g_lin[4], 0, wd[3])
            g2 = ops.cslice(c.g_rcat[d + 1], 0, wd[d])
            if d > 0:
                bn = self.down_bn[d].bn
                ops.bnact_bwd(c.e[d], None, c.g_lin[d + 1], c.g_e[d], g2=g2, bn=c.st_down[d], gamma=bn.weight.data,
                              beta=bn.bias.data, bn_eval=not c.train, act=ACT_LRELU, act2=ACT_RELU,
                              dgamma=bn.weight.grad if wgrad else None, dbeta=bn.bias.grad if wgrad else None)
            else:
                # (x is not read for its values here: the activation derivatives come from the saved output lin[1])
                ops.bnact_bwd(c.lin[1], c.lin[1], c.g_lin[1], c.g_e[0], g2=g2, act=ACT_LRELU, act2=ACT_RELU)
            if wgrad:
                weight_gradient(self.down[d], c.lin[d] if d > 0 else c.x_in, c.g_e[d], 2 * D - 1 - d)
            if d > 0:
                self.down[d].backward_data(c.g_e[d], c.g_lin[d])
        wg.flush('down')
        ops.SideStream.get(self.device).join()


# ------------------------------------------------------------------------------------------------
# PatchGAN discriminators (plain and selective-activation / masked)
# ------------------------------------------------------------------------------------------------
class PatchGANEngine:
    def __init__(self, module, masked, threshold, device, n_layers=3):
        self.module, self.masked, self.tau, self.device = module, masked, float(threshold), device
        seq = module.model
        idx = []
        if not masked:
            idx.append((0, None, None))
            i = 2
            for _ in range(n_layers):
                idx.append((i, i + 1, None))
                i += 3
            idx.append((i, None, None))
        else:
            idx.append((0, None, 2))
            i = 3
            for _ in range(n_layers):
                idx.append((i, i + 1, i + 2))
                i += 4
            idx.append((i, None, None))
        self.idx = idx
        self.L = len(idx)
        self.conv, self.bn, self.gate, self.inorm = [], [], [], []
        for li, (ci, bi, gi) in enumerate(idx):
            m = getattr(seq, str(ci))
            stride = 2 if li < self.L - 2 else 1
            self.conv.append(ConvOp(m.weight, m.bias, 4, stride, 1, False))
            norm = getattr(seq, str(bi)) if bi is not None else None
            # CycleGAN's plain discriminator normalises per image (InstanceNorm2d, no parameters): models/CycleGAN.py:139-177
            self.inorm.append(isinstance(norm, nn.InstanceNorm2d))
            self.bn.append(BNOp(norm) if isinstance(norm, nn.BatchNorm2d) else None)
            self.gate.append(getattr(seq, str(gi)) if gi is not None else None)
        self.chan = [c.rows for c in self.conv]
        self.in_nc = self.conv[0].cols
        self.mask = [torch.ones(self.chan[i], dtype=torch.float32, device=device) if self.gate[i] is not None else None
                     for i in range(self.L)]
        self.hook_layers = (1, 3)      # BN of layer 1 ('model.3'/'model.4') and layer 3 ('model.9'/'model.12')
        self.ctx = {}
        self.gbuf = {}

    def convs(self):
        return self.conv

    def grad_segments(self):
        """weight-optimizer parameters in the order backward() completes their gradients: the last conv first, then each
        layer's BatchNorm + conv down to the first conv (alphas belong to the arch optimizer)"""
        segs = []
        for li in range(self.L - 1, -1, -1):
            seg = []
            if self.bn[li] is not None:
                seg += list(self.bn[li].bn.parameters())
            seg += [p for p in (self.conv[li].weight, self.conv[li].bias) if p is not None]
            segs.append(seg)
        return segs

    def _seg_done(self, i):
        r = getattr(self, 'reducer', None)
        if r is not None and getattr(self, 'reduce_now', False):
            r.segment_done(i, ops.SideStream.get(self.device).stream if OVERLAP_WGRAD else None)

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.conv, self.device)
        self._pack.run()

    def refresh_masks(self):
        refresh_gate_masks(self)

    def _sizes(self, H, W):
        hs = []
        h, w = H, W
        for li in range(self.L):
            s = self.conv[li].stride
            h, w = (h + 2 - 4) // s + 1, (w + 2 - 4) // s + 1
            hs.append((h, w))
        return hs

    def new_ctx(self, N, H, W, tag):
        key = (N, H, W, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev = self.device
        c = type('DCtx', (), {})()
        c.N, c.H, c.W = N, H, W
        c.hs = self._sizes(H, W)
        c.x_in = ops.new_act(N, self.in_nc, H, W, dev)
        c.a0 = ops.new_act(N, self.chan[0], c.hs[0][0], c.hs[0][1], dev)
        c.g0 = ops.new_act(N, self.chan[0], c.hs[0][0], c.hs[0][1], dev) if self.masked else c.a0
        c.c = [None] * self.L
        c.y = [None] * self.L
        c.st = [None] * self.L
        for li in range(1, self.L - 1):
            c.c[li] = ops.new_act(N, self.chan[li], c.hs[li][0], c.hs[li][1], dev)
            c.y[li] = ops.new_act(N, self.chan[li], c.hs[li][0], c.hs[li][1], dev)
            c.st[li] = ops.INState(N, self.chan[li], dev) if self.inorm[li] else ops.BNState(self.chan[li], dev)
        c.pred = ops.new_act(N, 1, c.hs[-1][0], c.hs[-1][1], dev)
        self.ctx[key] = c
        return c

    def _gbufs(self, N, H, W, slot=0):
        key = (N, H, W) if slot == 0 else (N, H, W, slot)       # slot: a second set for a backward pass that runs beside another
        if key not in self.gbuf:
            hs = self._sizes(H, W)
            dev = self.device
            g = type('DGrad', (), {})()
            g.layer = [ops.new_act(N, self.chan[li], hs[li][0], hs[li][1], dev) for li in range(self.L)]
            g.x_in = ops.new_act(N, self.in_nc, H, W, dev)
            self.gbuf[key] = g
        return self.gbuf[key]

    def features(self, c):
        return [c.y[1], c.y[3]]

    def grad_pred_buffer(self, c, slot=0):
        return self._gbufs(c.N, c.H, c.W, slot).layer[-1]

    # ---------------------------------------------------------------------------------------
    def forward(self, c, train=True, defer_running=False, refresh=True):
        """input already in c.x_in; returns c.pred.
        defer_running: this pass runs ahead of its place in the reference's order (on another stream): its BatchNorm
        layers compute their coefficients but leave running_mean / running_var alone; apply_deferred_running(c) replays
        the finalize with the running update where the pass belongs (the EMA updates of two passes do not commute).
        refresh=False: the gate masks are current (refresh_masks() was called since alpha last changed)."""
        L = self.L
        if self.masked and refresh:
            self.refresh_masks()
        c.deferred = []
        if self.masked:         # the first gate (applied after the LeakyReLU, models/Pix2Pix.py:320-322) by the conv launch itself
            self.conv[0].forward(c.x_in, c.a0, act=ACT_LRELU, y2=c.g0, y2_mode=ops.Y2_GATE, y2_gate=self.mask[0])
        else:
            self.conv[0].forward(c.x_in, c.a0, act=ACT_LRELU)
        src = c.g0
        for li in range(1, L - 1):
            if self.inorm[li]:
                self.conv[li].forward(src, c.c[li])
                if c.hs[li][0] * c.hs[li][1] <= ops.INORM_FUSED_MAX_HW:      # statistics + normalisation in one launch
                    ops.inorm_fwd(c.c[li], c.y[li], c.st[li], act=ACT_LRELU)
                else:
                    ops.in_finalize(ops.channel_stats(c.c[li]), c.hs[li][0] * c.hs[li][1], c.st[li])
                    ops.bnact_fwd(c.c[li], c.y[li], scale=c.st[li].scale, shift=c.st[li].shift, act=ACT_LRELU, groups=c.N)
            else:
                n = c.N * c.hs[li][0] * c.hs[li][1]
                if defer_running and train:
                    # coefficients now (by the conv launch itself), running statistics when the pass's place comes
                    _, stats = self.conv[li].forward(src, c.c[li], want_stats=True,
                                                     bn=ops.bn_desc(self.bn[li].bn, c.st[li], n, self.device, running=False))
                    c.deferred.append((li, stats, n))
                elif train:
                    self.conv[li].forward(src, c.c[li], want_stats=True, bn=self.bn[li].desc(c.st[li], n, self.device))
                else:
                    _, stats = self.conv[li].forward(src, c.c[li], want_stats=True)
                    self.bn[li].finalize(stats, n, c.st[li], train)
                ops.bnact_fwd(c.c[li], c.y[li], scale=c.st[li].scale, shift=c.st[li].shift, gate=self.mask[li], act=ACT_LRELU)
            src = c.y[li]
        self.conv[L - 1].forward(src, c.pred)
        c.train = train
        return c.pred

    def apply_deferred_running(self, c):
        """the running-statistics updates a defer_running pass left out, on the current stream (which has joined the
        stream that pass ran on): the same finalize over the same partial sums, now with the EMA update"""
        for li, stats, n in getattr(c, 'deferred', []):
            stats.record_stream(ops.current_stream())
            self.bn[li].finalize(stats, n, c.st[li], True)
        c.deferred = []

    def backward(self, c, has_pred_grad=True, g_feat=None, wgrad=True, agrad=False, need_dx=True, gslot=0, dalpha=None):
        """dL/dpred must be in grad_pred_buffer(c) when has_pred_grad; g_feat = optional [g(y1), g(y3)].
        wgrad: accumulate conv/BN parameter gradients; agrad: accumulate alpha gradients.
        gslot / dalpha: a pass that runs beside another one of this discriminator (the architecture step's second pass, on the
        auxiliary stream) works in its own gradient buffers and accumulates its alpha gradients into dalpha[layer] instead of
        alpha.grad (the caller folds them in afterwards, in the reference's order).
        Returns dL/d(x_in) (NHWC bf16, same layout as c.x_in) when need_dx."""
        L = self.L
        G = self._gbufs(c.N, c.H, c.W, gslot)
        assert gslot == 0 or not wgrad

        def agrad_of(li):
            if not agrad or self.gate[li] is None:
                return None
            return dalpha[li] if dalpha is not None else self.gate[li].alpha.grad
        if g_feat is None:
            g_feat = [None, None]
        feat_of = {1: g_feat[0], 3: g_feat[1]}
        # (layers small enough for a grouped weight gradient -- the batch-1 models' discriminators -- are collected and run at the
        # end of the pass; the headline configuration's layers are far above ConvOp.group_entry's size rule and launch at once)
        wg = WgradCollector(self, c)
        if has_pred_grad:
            if wgrad:
                wg.add(self.conv[L - 1], c.y[L - 2], G.layer[L - 1], seg=0)
            self.conv[L - 1].backward_data(G.layer[L - 1], G.layer[L - 2])
        for li in range(L - 2, 0, -1):
            g1, g2 = G.layer[li], feat_of.get(li)
            if li == L - 2 and not has_pred_grad:
                assert g2 is not None, 'nothing to back-propagate'
                g1, g2 = g2, None
            gate = self.gate[li]
            if self.inorm[li]:
                if g2 is None and c.hs[li][0] * c.hs[li][1] <= ops.INORM_FUSED_MAX_HW:
                    ops.inorm_bwd(c.c[li], c.y[li], g1, G.layer[li], c.st[li], act=ACT_LRELU)
                else:
                    ops.bnact_bwd(c.c[li], c.y[li], g1, G.layer[li], g2=g2, bn=c.st[li], act=ACT_LRELU, act2=ACT_LRELU, groups=c.N)
            else:
                bn = self.bn[li].bn
                ops.bnact_bwd(c.c[li], None, g1, G.layer[li], g2=g2, bn=c.st[li], gamma=bn.weight.data, beta=bn.bias.data,
                              gate=self.mask[li], act=ACT_LRELU, act2=ACT_LRELU, dgamma=bn.weight.grad if wgrad else None,
                              dbeta=bn.bias.grad if wgrad else None,
                              dalpha=agrad_of(li))
            src = c.g0 if li == 1 else c.y[li - 1]
            if wgrad:
                wg.add(self.conv[li], src, G.layer[li], seg=L - 1 - li)
            self.conv[li].backward_data(G.layer[li], G.layer[li - 1])
        gate = self.gate[0]
        # the first conv's bias gradient is the channel sum of dz, which this pass forms anyway (its 'dbeta' sum): no separate
        # read of the [N, ndf, H/2, W/2] gradient by gcc_channel_sum (67 MB per discriminator pass at ndf 128, 256 x 256, N = 16)
        b0 = self.conv[0].bias
        ops.bnact_bwd(c.a0, None, G.layer[0], G.layer[0], gate=self.mask[0], gate_after_act=True, in_act=ACT_LRELU,
                      dalpha=agrad_of(0),
                      dbeta=b0.grad if (wgrad and b0 is not None) else None)
        if wgrad:
            wg.add(self.conv[0], c.x_in, G.layer[0], seg=L - 1, bias_done=True)
        dx = None
        if need_dx:
            self.conv[0].backward_data(G.layer[0], G.x_in)
            dx = G.x_in
        wg.flush('all')
        ops.SideStream.get(self.device).join()
        return dx


# ------------------------------------------------------------------------------------------------
# MobileResnet generator (--backbone resnet; CycleGAN)
# ------------------------------------------------------------------------------------------------
class DWOp:
    """depthwise 3x3 conv behind a ReflectionPad2d(1): fp32 master [C,1,3,3] + bias read directly by the kernel"""

    def __init__(self, conv):
        self.weight, self.bias = conv.weight, conv.bias
        self.C = conv.weight.shape[0]

    def forward(self, x, out):
        ops.dwconv_fwd(x, self.weight.data, self.bias.data if self.bias is not None else None, out)

    def backward_data(self, dy, out):
        ops.dwconv_bwd_data(dy, self.weight.data, out)

    def backward_weight(self, x, dy):
        def run():
            ops.dwconv_wgrad(x, dy, self.weight.grad, self.bias.grad if self.bias is not None else None)
        if OVERLAP_WGRAD:
            side = ops.SideStream.get(x.device)
            side.fork()
            with ops.on_stream(side.stream):
                run()
        else:
            run()


class MobileResnetEngine:
    """Forward / backward schedule of MobileResnetGenerator (models/Pix2Pix.py:132-265, models/CycleGAN.py:77-138):
    pad3+conv7 | 2x conv3 s2 | residual blocks of (pad1, dw3x3, IN, 1x1, IN, ReLU, pad1, dw3x3, IN, 1x1, IN) | 2x convT3 s2 |
    pad3+conv7+tanh, InstanceNorm2d(affine=False) after every conv.  Widths come from the module tree (pruned cfgs
    with removed blocks included: a removed block is simply absent from the Sequential)."""

    def __init__(self, module, device):
        self.module, self.device = module, device
        items = list(module.model.named_children())
        convs = [(int(n), m) for n, m in items if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
        blocks = [(int(n), m) for n, m in items if hasattr(m, 'conv_block')]
        assert len(convs) == 6 and not isinstance(convs[0][1], nn.ConvTranspose2d)
        self.stem = [ConvOp(convs[0][1].weight, convs[0][1].bias, 7, 1, 0, False),
                     ConvOp(convs[1][1].weight, convs[1][1].bias, 3, 2, 1, False),
                     ConvOp(convs[2][1].weight, convs[2][1].bias, 3, 2, 1, False)]
        self.ups = [ConvOp(convs[3][1].weight, convs[3][1].bias, 3, 2, 1, True),
                    ConvOp(convs[4][1].weight, convs[4][1].bias, 3, 2, 1, True)]
        self.last = ConvOp(convs[5][1].weight, convs[5][1].bias, 7, 1, 0, False)
        self.blocks = []
        for idx, blk in blocks:
            cb = blk.conv_block
            s1, s2 = getattr(cb, '1').conv, getattr(cb, '6').conv
            self.blocks.append(type('Blk', (), dict(
                index=idx, dw1=DWOp(getattr(s1, '0')), pw1=ConvOp(getattr(s1, '2').weight, getattr(s1, '2').bias, 1, 1, 0, False),
                dw2=DWOp(getattr(s2, '0')), pw2=ConvOp(getattr(s2, '2').weight, getattr(s2, '2').bias, 1, 1, 0, False)))())
        self.relu_index = convs[2][0] + 2          # 'model.9': the ReLU behind the second stride-2 conv
        self.in_nc, self.out_nc = self.stem[0].cols, self.last.rows
        self.hook_names = ['model.9', 'model.12', 'model.15', 'model.18']
        self.ctx, self.gbuf = {}, {}

    def convs(self):
        return self.stem + [c for b in self.blocks for c in (b.pw1, b.pw2)] + self.ups + [self.last]

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.convs(), self.device)
        self._pack.run()

    # ---------------------------------------------------------------------------------------
    def _ctx(self, N, H, W, tag='main'):
        key = (N, H, W, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev = self.device
        c = type('ResnetCtx', (), {})()
        c.N, c.H, c.W = N, H, W
        new = lambda Cc, h, w: ops.new_act(N, Cc, h, w, dev)
        c.x_in = new(self.in_nc, H, W)
        c.xpad = new(self.in_nc, H + 6, W + 6)
        sz = [(H, W), (H // 2, W // 2), (H // 4, W // 4)]
        c.sz = sz
        c.s_raw = [new(self.stem[i].rows, *sz[i]) for i in range(3)]
        c.s_act = [new(self.stem[i].rows, *sz[i]) for i in range(3)]
        c.s_st = [ops.INState(N, self.stem[i].rows, dev) for i in range(3)]
        h, w = sz[2]
        c.blk = []
        for b in self.blocks:
            t = type('BlkCtx', (), {})()
            cin, cmid, cout = b.dw1.C, b.pw1.rows, b.pw2.rows
            t.d1, t.n1, t.p1, t.r1 = new(cin, h, w), new(cin, h, w), new(cmid, h, w), new(cmid, h, w)
            t.d2, t.n2, t.p2, t.o = new(cmid, h, w), new(cmid, h, w), new(cout, h, w), new(cout, h, w)
            t.st = [ops.INState(N, cc, dev) for cc in (cin, cmid, cmid, cout)]
            c.blk.append(t)
        usz = [sz[1], sz[0]]
        c.t_raw = [new(self.ups[i].cols, *usz[i]) for i in range(2)]
        c.t_act = [new(self.ups[i].cols, *usz[i]) for i in range(2)]
        c.t_st = [ops.INState(N, self.ups[i].cols, dev) for i in range(2)]
        c.ypad = new(self.ups[1].cols, H + 6, W + 6)
        c.out = new(self.out_nc, H, W)
        c.g_out = new(self.out_nc, H, W)
        self.ctx[key] = c
        return c

    def _gbufs(self, N, H, W):
        """gradient buffers shared by all contexts of one shape (a backward joins the side stream before returning)"""
        key = (N, H, W)
        if key in self.gbuf:
            return self.gbuf[key]
        dev = self.device
        new = lambda Cc, h, w: ops.new_act(N, Cc, h, w, dev)
        g = type('ResnetGrad', (), {})()
        sz = [(H, W), (H // 2, W // 2), (H // 4, W // 4)]
        h, w = sz[2]
        g.ypad = new(self.ups[1].cols, H + 6, W + 6)
        g.t = [new(self.ups[0].cols, *sz[1]), new(self.ups[1].cols, *sz[0])]
        g.h = new(self.stem[2].rows, h, w)
        g.tmp = new(self.stem[2].rows, h, w)
        g.blk = []
        for b in self.blocks:
            cin, cmid, cout = b.dw1.C, b.pw1.rows, b.pw2.rows
            g.blk.append((new(cout, h, w), new(cmid, h, w), new(cmid, h, w), new(cin, h, w)))
        g.s = [new(self.stem[i].rows, *sz[i]) for i in range(2)]
        g.xpad = new(self.in_nc, H + 6, W + 6)
        g.x_in = new(self.in_nc, H, W)
        self.gbuf[key] = g
        return g

    def features(self, c):
        """outputs of the hooked modules ('model.9' ReLU, blocks 'model.12/15/18'), in that order"""
        out = []
        by_index = {b.index: i for i, b in enumerate(self.blocks)}
        for name in self.hook_names:
            i = int(name.split('.')[1])
            if i == self.relu_index:
                out.append(c.s_act[2])
            elif i in by_index:
                out.append(c.blk[by_index[i]].o)
            else:
                raise NotImplementedError('distillation hook %s is not a residual block of this (pruned) generator' % name)
        return out

    @staticmethod
    def _inorm(x, y, st, act=ACT_NONE, residual=None):
        N, _, H, W = x.shape
        if H * W <= ops.INORM_FUSED_MAX_HW:
            ops.inorm_fwd(x, y, st, act=act, residual=residual)
            return
        ops.in_finalize(ops.channel_stats(x), H * W, st)
        ops.bnact_fwd(x, y, scale=st.scale, shift=st.shift, act=act, groups=N, residual=residual)

    @staticmethod
    def _inorm_bwd(x, y, g, dx, st, act=ACT_NONE):
        N, _, H, W = x.shape
        if H * W <= ops.INORM_FUSED_MAX_HW:
            ops.inorm_bwd(x, y, g, dx, st, act=act)
        else:
            ops.bnact_bwd(x, y, g, dx, bn=st, act=act, groups=N)

    # ---------------------------------------------------------------------------------------
    def forward(self, c, train=True):
        """input already in c.x_in; returns c (c.out = tanh image).  InstanceNorm without running statistics and
        dropout rate 0: train and eval are the same arithmetic."""
        ops.reflect_pad(c.x_in, c.xpad, 3)
        src = c.xpad
        for i in range(3):
            self.stem[i].forward(src, c.s_raw[i])
            self._inorm(c.s_raw[i], c.s_act[i], c.s_st[i], act=ACT_RELU)
            src = c.s_act[i]
        for b, t in zip(self.blocks, c.blk):
            b.dw1.forward(src, t.d1)
            self._inorm(t.d1, t.n1, t.st[0])
            b.pw1.forward(t.n1, t.p1)
            self._inorm(t.p1, t.r1, t.st[1], act=ACT_RELU)
            b.dw2.forward(t.r1, t.d2)
            self._inorm(t.d2, t.n2, t.st[2])
            b.pw2.forward(t.n2, t.p2)
            self._inorm(t.p2, t.o, t.st[3], residual=src)
            src = t.o
        for i in range(2):
            self.ups[i].forward(src, c.t_raw[i])
            self._inorm(c.t_raw[i], c.t_act[i], c.t_st[i], act=ACT_RELU)
            src = c.t_act[i]
        ops.reflect_pad(src, c.ypad, 3)
        self.last.forward(c.ypad, c.out, act=ACT_TANH)
        return c

    # ---------------------------------------------------------------------------------------
    def backward(self, c, g_feat=None, wgrad=True, need_dx=False):
        """c.g_out holds dL/d(out); g_feat: optional gradients w.r.t. features(c).  Accumulates parameter gradients
        when wgrad; returns dL/d(x_in) when need_dx (the cycle / identity chains of CycleGAN)."""
        N = c.N
        G = self._gbufs(c.N, c.H, c.W)
        gf = {}
        if g_feat is not None:
            for name, g in zip(self.hook_names, g_feat):
                if g is not None:
                    gf[int(name.split('.')[1])] = g
        ops.bnact_bwd(c.out, None, c.g_out, c.g_out, in_act=ACT_TANH)
        # the ConvOp layers' weight gradients (k7 / k3 s2 / ConvTranspose / the 18 pointwise convs) are collected and run as one
        # grouped launch at the end of the pass (WgradCollector); the depthwise layers keep their own kernel
        wg = WgradCollector(self, c)
        if wgrad:
            wg.add(self.last, c.ypad, c.g_out)
        self.last.backward_data(c.g_out, G.ypad)
        ops.reflect_pad(G.ypad, G.t[1], 3, backward=True)
        n_blk = len(self.blocks)
        for i in (1, 0):
            self._inorm_bwd(c.t_raw[i], c.t_act[i], G.t[i], G.t[i], c.t_st[i], act=ACT_RELU)
            src = c.t_act[0] if i == 1 else (c.blk[-1].o if n_blk else c.s_act[2])
            if wgrad:
                wg.add(self.ups[i], src, G.t[i])
            self.ups[i].backward_data(G.t[i], G.t[0] if i == 1 else G.h)
        for bi in range(n_blk - 1, -1, -1):
            b, t = self.blocks[bi], c.blk[bi]
            gp2, gn2, gr1, gn1 = G.blk[bi]
            x_in = c.blk[bi - 1].o if bi > 0 else c.s_act[2]
            if b.index in gf:
                ops.nhwc_add(gf[b.index], 0, G.h, 0, G.h.shape[1])
            self._inorm_bwd(t.p2, None, G.h, gp2, t.st[3])
            if wgrad:
                wg.add(b.pw2, t.n2, gp2)
            b.pw2.backward_data(gp2, gn2)
            self._inorm_bwd(t.d2, None, gn2, gn2, t.st[2])
            if wgrad:
                b.dw2.backward_weight(t.r1, gn2)
            b.dw2.backward_data(gn2, gr1)
            self._inorm_bwd(t.p1, t.r1, gr1, gr1, t.st[1], act=ACT_RELU)
            if wgrad:
                wg.add(b.pw1, t.n1, gr1)
            b.pw1.backward_data(gr1, gn1)
            self._inorm_bwd(t.d1, None, gn1, gn1, t.st[0])
            if wgrad:
                b.dw1.backward_weight(x_in, gn1)
            b.dw1.backward_data(gn1, G.tmp)
            ops.nhwc_add(G.tmp, 0, G.h, 0, G.h.shape[1])
            if (n_blk - bi) % 3 == 0:
                wg.flush('blk%d' % bi)             # a group per three blocks, beside the next blocks' chain
        if self.relu_index in gf:
            ops.nhwc_add(gf[self.relu_index], 0, G.h, 0, G.h.shape[1])
        g = G.h
        for i in (2, 1, 0):
            self._inorm_bwd(c.s_raw[i], c.s_act[i], g, g, c.s_st[i], act=ACT_RELU)
            if wgrad:
                wg.add(self.stem[i], c.s_act[i - 1] if i > 0 else c.xpad, g)
            if i > 0:
                self.stem[i].backward_data(g, G.s[i - 1])
                g = G.s[i - 1]
        dx = None
        if need_dx:
            self.stem[0].backward_data(g, G.xpad)
            ops.reflect_pad(G.xpad, G.x_in, 3, backward=True)
            dx = G.x_in
        wg.flush('all')
        ops.SideStream.get(self.device).join()
        return dx


# ------------------------------------------------------------------------------------------------
# SAGAN: spectrally normalised convs, self attention, generator / discriminator engines
# ------------------------------------------------------------------------------------------------
class SNState:
    """what one forward call of a spectrally normalised conv leaves behind for its backward: the bf16 packings of
    W_bar / sigma of THAT call, sigma, and t = W_bar v (gradient of u)"""

    def __init__(self, op):
        dev = op.w_bar.device
        taps = op.k * op.k
        self.w = torch.zeros((ops.ceil8(op.rows), taps, ops.ceil8(op.cols)), dtype=torch.bfloat16, device=dev)
        self.wt = torch.zeros((ops.ceil8(op.cols), taps, ops.ceil8(op.rows)), dtype=torch.bfloat16, device=dev)
        self.sigma = torch.zeros(1, dtype=torch.float32, device=dev)
        self.t = torch.zeros(op.rows, dtype=torch.float32, device=dev)


class SNConvOp:
    """SpectralNorm(Conv2d | ConvTranspose2d) (models/SAGAN.py:17-70): parameters weight_bar, weight_u, weight_v, bias
    of the wrapped module.  Every forward runs one power iteration (it moves u, v: eval passes too) and convolves
    with W_bar / sigma; the weight gradient is folded through sigma into weight_bar.grad (and, for the discriminator
    whose u, v the reference trains, into their gradients)."""

    def __init__(self, inner, k, stride, pad, transposed, train_uv=False):
        self.w_bar, self.u, self.v, self.bias = inner.weight_bar, inner.weight_u, inner.weight_v, inner.bias
        self.k, self.stride, self.pad, self.transposed, self.train_uv = k, stride, pad, transposed, train_uv
        self.rows, self.cols = self.w_bar.shape[0], self.w_bar.shape[1]
        d = self.w_bar.data
        self.w_eff = torch.empty_strided(d.shape, d.stride(), dtype=torch.float32, device=d.device)
        self.g_eff = torch.empty_strided(d.shape, d.stride(), dtype=torch.float32, device=d.device)

    def new_state(self):
        return SNState(self)

    def fused_pack_ok(self):
        return ops.SN_FUSED_PACK and self.w_bar.data.is_contiguous(memory_format=torch.channels_last if self.k > 1 else torch.contiguous_format)

    @staticmethod
    def iterate_all(pairs):
        """the power iterations of a forward pass's layers [(SNConvOp, SNState)] up front, grouped (they depend on the weights alone):
        four launches instead of four per layer; True when done -- the layers' forward(..., iterated=True) then skip their own"""
        if not SN_GROUP or len(pairs) < 2 or not all(op.fused_pack_ok() for op, _ in pairs):
            return False
        ops.spectral_power_iteration_pack_group([(op.w_bar.data, op.u.data, op.v.data, st.t, st.sigma, st.w, st.wt) for op, st in pairs])
        return True

    def forward(self, st, x, out, act=ACT_NONE, slope=LRELU, want_stats=False, bn=None, iterated=False):
        if iterated:
            pass
        elif self.fused_pack_ok():
            ops.spectral_power_iteration_pack(self.w_bar.data, self.u.data, self.v.data, st.t, st.sigma, st.w, st.wt)
        else:
            ops.spectral_power_iteration(self.w_bar.data, self.u.data, self.v.data, st.t, st.sigma, self.w_eff)
            ops.pack_weights_into(self.w_eff, st.w, st.wt)
        b = self.bias.data if self.bias is not None else None
        if not self.transposed:
            return ops.conv_fprop(x, st.w, self.rows, self.k, self.stride, self.pad, out=out, bias=b, act=act, slope=slope,
                                  want_stats=want_stats, bn=bn)
        _, _, H, W = out.shape
        return ops.conv_dgrad(x, st.wt, self.cols, H, W, self.k, self.stride, self.pad, out=out, bias=b, act=act, slope=slope,
                              want_stats=want_stats, bn=bn)

    def backward_data(self, st, dy, out):
        if not self.transposed:
            _, _, H, W = out.shape
            return ops.conv_dgrad(dy, st.wt, self.cols, H, W, self.k, self.stride, self.pad, out=out)
        return ops.conv_fprop(dy, st.w, self.rows, self.k, self.stride, self.pad, out=out)

    def backward_weight(self, st, x, dy):
        def run():
            cx, cdy = (x, dy) if not self.transposed else (dy, x)
            ops.conv_wgrad(cx, cdy, self.g_eff, self.k, self.stride, self.pad, accumulate=False)
            ops.spectral_grad(self.g_eff, self.w_bar.data, self.u.data, self.v.data, st.t, st.sigma, self.w_bar.grad,
                              du=self.u.grad if self.train_uv else None, dv=self.v.grad if self.train_uv else None)
            if self.bias is not None:
                ops.channel_sum(dy, self.bias.grad, accumulate=True)
        if OVERLAP_WGRAD:
            side = ops.SideStream.get(x.device)
            side.fork()
            with ops.on_stream(side.stream):
                run()
        else:
            run()


class AttnOp:
    """Self_Attn (models/SAGAN.py:72-104): three biased 1x1 convs write q | k | v into channel slices of one buffer,
    gcc_attention_* do the rest; y = gamma * attention + x"""

    def __init__(self, module, device):
        self.module = module
        q, k, v = module.query_conv, module.key_conv, module.value_conv
        self.C, self.C8 = v.weight.shape[0], q.weight.shape[0]
        self.convs = [ConvOp(m.weight, m.bias, 1, 1, 0, False) for m in (q, k, v)]
        c8p = ops.ceil8(self.C8)
        self.offs = (0, c8p, 2 * c8p)
        self.width = 2 * c8p + self.C
        self.slices = ((0, self.C8), (c8p, self.C8), (2 * c8p, self.C))
        self.device = device

    def new_state(self, N, H, W):
        st = type('AttnState', (), {})()
        dev = self.device
        st.qkv = ops.new_act(N, self.width, H, W, dev)
        st.o = ops.new_act(N, self.C, H, W, dev)
        st.y = ops.new_act(N, self.C, H, W, dev)
        st.stats = torch.zeros((N, H * W, 2), dtype=torch.float32, device=dev)
        return st

    def grad_buffers(self, N, H, W):
        g = type('AttnGrad', (), {})()
        dev = self.device
        g.dqkv = ops.new_act(N, self.width, H, W, dev)
        g.rowdot = torch.zeros((N, H * W), dtype=torch.float32, device=dev)
        g.tmp = ops.new_act(N, self.C, H, W, dev)
        return g

    def forward(self, st, x):
        for conv, (off, w) in zip(self.convs, self.slices):
            conv.forward(x, ops.cslice(st.qkv, off, w))
        ops.attention_fwd(st.qkv, self.offs, x, self.module.gamma.data, self.C, self.C8, st.y, st.o, st.stats)
        return st.y

    def backward(self, st, G, x, dy, dx, wgrad=True):
        """dy: gradient w.r.t. y; dx receives dy (residual branch) + the data gradients of the three 1x1 convs"""
        ops.attention_bwd(st.qkv, self.offs, st.o, st.stats, self.module.gamma.data, dy, self.C, self.C8, G.dqkv, G.rowdot,
                          dgamma=self.module.gamma.grad if wgrad else None)
        ops.nhwc_copy(dy, 0, dx, 0, self.C)
        for conv, (off, w) in zip(self.convs, self.slices):
            d = ops.cslice(G.dqkv, off, w)
            if wgrad:
                conv.backward_weight(x, d)
            conv.backward_data(d, G.tmp)
            ops.nhwc_add(G.tmp, 0, dx, 0, self.C)


class SaganGeneratorEngine:
    """Generator(image_size=64) (models/SAGAN.py:106-170): z -> 4x4 -> 8 -> 16 -> attn1 -> 32 -> attn2 -> 64x64"""

    def __init__(self, module, device):
        self.module, self.device = module, device
        geom = ((1, 0), (2, 1), (2, 1), (2, 1))
        self.sn = [SNConvOp(getattr(module, 'l%d' % (i + 1))[0].module, 4, s, p, True) for i, (s, p) in enumerate(geom)]
        self.bn = [BNOp(getattr(module, 'l%d' % (i + 1))[1]) for i in range(4)]
        self.last = ConvOp(module.last[0].weight, module.last[0].bias, 4, 2, 1, True)
        self.attn = [AttnOp(module.attn1, device), AttnOp(module.attn2, device)]
        self.width = [op.cols for op in self.sn]
        self.z_dim = self.sn[0].rows
        self.ctx, self.gbuf = {}, {}

    def convs(self):
        return [self.last] + [c for a in self.attn for c in a.convs]

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.convs(), self.device)
        self._pack.run()

    def _ctx(self, N, tag='main'):
        key = (N, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev = self.device
        c = type('SaganGCtx', (), {})()
        c.N = N
        c.z = ops.new_act(N, self.z_dim, 1, 1, dev)
        c.size = [4, 8, 16, 32]
        c.raw = [ops.new_act(N, self.width[i], c.size[i], c.size[i], dev) for i in range(4)]
        c.act = [ops.new_act(N, self.width[i], c.size[i], c.size[i], dev) for i in range(4)]
        c.bn = [ops.BNState(self.width[i], dev) for i in range(4)]
        c.sn = [op.new_state() for op in self.sn]
        c.attn = [self.attn[0].new_state(N, 16, 16), self.attn[1].new_state(N, 32, 32)]
        c.out = ops.new_act(N, 3, 64, 64, dev)
        c.g_out = ops.new_act(N, 3, 64, 64, dev)
        c.train = True
        self.ctx[key] = c
        return c

    def _gbufs(self, N):
        if N not in self.gbuf:
            dev = self.device
            g = type('SaganGGrad', (), {})()
            g.y = [ops.new_act(N, self.width[2], 16, 16, dev), ops.new_act(N, self.width[3], 32, 32, dev)]
            g.act = [ops.new_act(N, self.width[i], s, s, dev) for i, s in enumerate((4, 8, 16, 32))]
            g.raw = [ops.new_act(N, self.width[i], s, s, dev) for i, s in enumerate((4, 8, 16, 32))]
            g.attn = [self.attn[0].grad_buffers(N, 16, 16), self.attn[1].grad_buffers(N, 32, 32)]
            self.gbuf[N] = g
        return self.gbuf[N]

    def features(self, c):
        """hooks 'l2' (post-ReLU) and 'attn2'"""
        return [c.act[1], c.attn[1].y]

    def forward(self, c, train=True):
        """z already in c.z (NHWC [N,1,1,z_dim]); returns c (c.out = tanh image)"""
        c.train = train
        src = c.z
        it = SNConvOp.iterate_all([(self.sn[i], c.sn[i]) for i in range(4)])
        for i in range(4):
            n = c.N * c.size[i] * c.size[i]
            if train:
                self.sn[i].forward(c.sn[i], src, c.raw[i], want_stats=True, bn=self.bn[i].desc(c.bn[i], n, self.device), iterated=it)
            else:
                _, stats = self.sn[i].forward(c.sn[i], src, c.raw[i], want_stats=True, iterated=it)
                self.bn[i].finalize(stats, n, c.bn[i], train)
            ops.bnact_fwd(c.raw[i], c.act[i], scale=c.bn[i].scale, shift=c.bn[i].shift, act=ACT_RELU)
            src = c.act[i]
            if i >= 2:
                src = self.attn[i - 2].forward(c.attn[i - 2], src)
        self.last.forward(src, c.out, act=ACT_TANH)
        return c

    def backward(self, c, g_feat=None, wgrad=True):
        """c.g_out holds dL/d(image); g_feat: optional gradients w.r.t. features(c)"""
        G = self._gbufs(c.N)
        g_feat = g_feat or [None, None]
        ops.bnact_bwd(c.out, None, c.g_out, c.g_out, in_act=ACT_TANH)
        if wgrad:
            self.last.backward_weight(c.attn[1].y, c.g_out)
        self.last.backward_data(c.g_out, G.y[1])
        if g_feat[1] is not None:
            ops.nhwc_add(g_feat[1], 0, G.y[1], 0, self.width[3])
        for i in (3, 2, 1, 0):
            if i >= 2:
                self.attn[i - 2].backward(c.attn[i - 2], G.attn[i - 2], c.act[i], G.y[i - 2], G.act[i], wgrad=wgrad)
            if i == 1 and g_feat[0] is not None:
                ops.nhwc_add(g_feat[0], 0, G.act[1], 0, self.width[1])
            bn = self.bn[i].bn
            ops.bnact_bwd(c.raw[i], c.act[i], G.act[i], G.raw[i], bn=c.bn[i], gamma=bn.weight.data, beta=bn.bias.data,
                          bn_eval=not c.train, act=ACT_RELU, dgamma=bn.weight.grad if wgrad else None,
                          dbeta=bn.bias.grad if wgrad else None)
            src = c.z if i == 0 else (c.act[i - 1] if i - 1 < 2 else c.attn[i - 3].y)
            if wgrad:
                self.sn[i].backward_weight(c.sn[i], src, G.raw[i])
            if i > 0:
                self.sn[i].backward_data(c.sn[i], G.raw[i], G.y[i - 3] if i - 1 >= 2 else G.act[i - 1])
        ops.SideStream.get(self.device).join()


class SaganDiscriminatorEngine:
    """Discriminator / MaskDiscriminator (models/SAGAN.py:172-274): 4 x [SN conv k4 s2 (+gate) + LeakyReLU(0.1)], attention
    after l3 and l4, conv k4 on the 4x4 map -> one logit per image"""
    SLOPE = 0.1

    def __init__(self, module, masked, threshold, device):
        self.module, self.masked, self.tau, self.device = module, masked, float(threshold), device
        self.sn = [SNConvOp(getattr(module, 'l%d' % (i + 1))[0].module, 4, 2, 1, False, train_uv=True) for i in range(4)]
        self.gate = [getattr(module, 'l%d' % (i + 1))[1] if masked else None for i in range(4)]
        self.last = ConvOp(module.last[0].weight, module.last[0].bias, 4, 1, 0, False)
        self.attn = [AttnOp(module.attn1, device), AttnOp(module.attn2, device)]
        self.width = [op.rows for op in self.sn]
        self.mask = [torch.ones(w, dtype=torch.float32, device=device) if masked else None for w in self.width]
        self.size = [32, 16, 8, 4]
        self.ctx, self.gbuf = {}, {}

    def convs(self):
        return [self.last] + [c for a in self.attn for c in a.convs]

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.convs(), self.device)
        self._pack.run()

    def refresh_masks(self):
        refresh_gate_masks(self)

    def new_ctx(self, N, tag):
        key = (N, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev = self.device
        c = type('SaganDCtx', (), {})()
        c.N = N
        c.x_in = ops.new_act(N, 3, 64, 64, dev)
        c.raw = [ops.new_act(N, self.width[i], self.size[i], self.size[i], dev) for i in range(4)]
        c.act = [ops.new_act(N, self.width[i], self.size[i], self.size[i], dev) for i in range(4)]
        c.sn = [op.new_state() for op in self.sn]
        c.attn = [self.attn[0].new_state(N, 8, 8), self.attn[1].new_state(N, 4, 4)]
        c.pred = ops.new_act(N, 1, 1, 1, dev)
        self.ctx[key] = c
        return c

    def _gbufs(self, N):
        if N not in self.gbuf:
            dev = self.device
            g = type('SaganDGrad', (), {})()
            g.pred = ops.new_act(N, 1, 1, 1, dev)
            g.y = [ops.new_act(N, self.width[2], 8, 8, dev), ops.new_act(N, self.width[3], 4, 4, dev)]
            g.act = [ops.new_act(N, self.width[i], self.size[i], self.size[i], dev) for i in range(4)]
            g.raw = [ops.new_act(N, self.width[i], self.size[i], self.size[i], dev) for i in range(4)]
            g.attn = [self.attn[0].grad_buffers(N, 8, 8), self.attn[1].grad_buffers(N, 4, 4)]
            g.x_in = ops.new_act(N, 3, 64, 64, dev)
            self.gbuf[N] = g
        return self.gbuf[N]

    def features(self, c):
        """hooks 'l2' (post-LeakyReLU) and 'attn2'"""
        return [c.act[1], c.attn[1].y]

    def grad_pred_buffer(self, c):
        return self._gbufs(c.N).pred

    def forward(self, c):
        if self.masked:
            self.refresh_masks()
        src = c.x_in
        it = SNConvOp.iterate_all([(self.sn[i], c.sn[i]) for i in range(4)])
        for i in range(4):
            self.sn[i].forward(c.sn[i], src, c.raw[i], iterated=it)
            ops.bnact_fwd(c.raw[i], c.act[i], gate=self.mask[i], act=ACT_LRELU, slope=self.SLOPE)
            src = c.act[i]
            if i >= 2:
                src = self.attn[i - 2].forward(c.attn[i - 2], src)
        self.last.forward(src, c.pred)
        return c.pred

    def backward(self, c, has_pred_grad=True, g_feat=None, wgrad=True, agrad=False, need_dx=True):
        """dL/dpred in grad_pred_buffer(c) when has_pred_grad; g_feat = optional [g('l2'), g('attn2')]"""
        G = self._gbufs(c.N)
        g_feat = g_feat or [None, None]
        if has_pred_grad:
            if wgrad:
                self.last.backward_weight(c.attn[1].y, G.pred)
            self.last.backward_data(G.pred, G.y[1])
            if g_feat[1] is not None:
                ops.nhwc_add(g_feat[1], 0, G.y[1], 0, self.width[3])
        else:
            assert g_feat[1] is not None, 'nothing to back-propagate'
            ops.nhwc_copy(g_feat[1], 0, G.y[1], 0, self.width[3])
        for i in (3, 2, 1, 0):
            if i >= 2:
                self.attn[i - 2].backward(c.attn[i - 2], G.attn[i - 2], c.act[i], G.y[i - 2], G.act[i], wgrad=wgrad)
            if i == 1 and g_feat[0] is not None:
                ops.nhwc_add(g_feat[0], 0, G.act[1], 0, self.width[1])
            gate = self.gate[i]
            ops.bnact_bwd(c.raw[i], c.act[i], G.act[i], G.raw[i], gate=self.mask[i], act=ACT_LRELU, slope=self.SLOPE,
                          dalpha=gate.alpha.grad if (agrad and gate is not None) else None)
            src = c.x_in if i == 0 else (c.act[i - 1] if i - 1 < 2 else c.attn[i - 3].y)
            if wgrad:
                self.sn[i].backward_weight(c.sn[i], src, G.raw[i])
            if i > 0:
                self.sn[i].backward_data(c.sn[i], G.raw[i], G.y[i - 3] if i - 1 >= 2 else G.act[i - 1])
            elif need_dx:
                self.sn[0].backward_data(c.sn[0], G.raw[0], G.x_in)
        ops.SideStream.get(self.device).join()
        return G.x_in if need_dx else None


# ------------------------------------------------------------------------------------------------
# SRGAN: SRResNet generator, avg-pool discriminator, truncated VGG19 (models/SRGAN.py, models/GANLoss.py:95-145)
# ------------------------------------------------------------------------------------------------
class SRResNetEngine:
    """Generator (models/SRGAN.py:139-199): conv9 + PReLU | n x [conv3 BN PReLU conv3 BN, + x] | conv3 BN + long skip |
    2 x [conv3 -> PixelShuffle(2) -> PReLU] | conv9 + tanh.  train_prelu False: the slopes receive no gradient (the
    reference's distillation optimizer does not hold them)."""

    def __init__(self, module, device, train_prelu=True):
        self.module, self.device, self.train_prelu = module, device, train_prelu
        cb = lambda m, i: getattr(m.conv_block, str(i))
        c1 = module.conv_block1
        self.first = ConvOp(cb(c1, 0).weight, cb(c1, 0).bias, 9, 1, 4, False)
        self.first_slope = cb(c1, 1).weight
        self.blocks = []
        for blk in module.residual_blocks.children():
            b1, b2 = blk.conv_block1, blk.conv_block2
            self.blocks.append(type('SRBlock', (), dict(
                conv1=ConvOp(cb(b1, 0).weight, cb(b1, 0).bias, 3, 1, 1, False), bn1=BNOp(cb(b1, 1)), slope=cb(b1, 2).weight,
                conv2=ConvOp(cb(b2, 0).weight, cb(b2, 0).bias, 3, 1, 1, False), bn2=BNOp(cb(b2, 1))))())
        c2 = module.conv_block2
        self.mid = ConvOp(cb(c2, 0).weight, cb(c2, 0).bias, 3, 1, 1, False)
        self.mid_bn = BNOp(cb(c2, 1))
        self.sub = [(ConvOp(s.conv.weight, s.conv.bias, 3, 1, 1, False), s.prelu.weight)
                    for s in module.subpixel_convolutional_blocks.children()]
        c3 = module.conv_block3
        self.last = ConvOp(cb(c3, 0).weight, cb(c3, 0).bias, 9, 1, 4, False)
        self.C = self.first.rows
        self.hook_blocks = (3, 7, 11, 15)
        self.ctx, self.gbuf = {}, {}

    def convs(self):
        return [self.first, self.mid, self.last] + [c for b in self.blocks for c in (b.conv1, b.conv2)] + [c for c, _ in self.sub]

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.convs(), self.device)
        self._pack.run()

    def _ctx(self, N, H, W, tag='main'):
        key = (N, H, W, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev, C = self.device, self.C
        new = lambda Cc, s=1: ops.new_act(N, Cc, H * s, W * s, dev)
        c = type('SRCtx', (), {})()
        c.N, c.H, c.W = N, H, W
        c.x_in = new(3)
        c.raw0, c.h0 = new(C), new(C)
        c.blk = []
        for b in self.blocks:
            t = type('SRBlkCtx', (), {})()
            ci = b.conv1.rows
            t.r1, t.z1, t.a1, t.r2, t.out = new(ci), new(ci), new(ci), new(C), new(C)
            t.st1, t.st2 = ops.BNState(ci, dev), ops.BNState(C, dev)
            c.blk.append(t)
        c.rm, c.hm, c.st_m = new(C), new(C), ops.BNState(C, dev)
        c.s_raw = [new(4 * C, 1), new(4 * C, 2)]
        c.s_act = [new(C, 2), new(C, 4)]
        c.out = new(3, 4)
        c.g_out = new(3, 4)
        c.train = True
        self.ctx[key] = c
        return c

    def _gbufs(self, N, H, W):
        key = (N, H, W)
        if key not in self.gbuf:
            dev, C = self.device, self.C
            new = lambda Cc, s=1: ops.new_act(N, Cc, H * s, W * s, dev)
            g = type('SRGrad', (), {})()
            g.s_act = [new(C, 2), new(C, 4)]
            g.s_raw = [new(4 * C, 1), new(4 * C, 2)]
            g.hm, g.rm, g.h, g.tmp, g.h0, g.raw0 = new(C), new(C), new(C), new(C), new(C), new(C)
            g.blk = [(new(C), new(b.conv1.rows), new(b.conv1.rows), new(b.conv1.rows)) for b in self.blocks]
            self.gbuf[key] = g
        return self.gbuf[key]

    def features(self, c):
        return [c.blk[i].out for i in self.hook_blocks if i < len(c.blk)]

    def forward(self, c, train=True):
        """low-resolution image already in c.x_in; returns c (c.out = tanh image at 4x)"""
        c.train = train
        N, hw = c.N, c.H * c.W
        self.first.forward(c.x_in, c.raw0)
        ops.prelu_fwd(c.raw0, self.first_slope.data, c.h0)
        h = c.h0
        for b, t in zip(self.blocks, c.blk):
            if train:
                b.conv1.forward(h, t.r1, want_stats=True, bn=b.bn1.desc(t.st1, N * hw, self.device))
            else:
                _, st = b.conv1.forward(h, t.r1, want_stats=True)
                b.bn1.finalize(st, N * hw, t.st1, train)
            ops.bnact_fwd(t.r1, t.z1, scale=t.st1.scale, shift=t.st1.shift)
            ops.prelu_fwd(t.z1, b.slope.data, t.a1)
            if train:
                b.conv2.forward(t.a1, t.r2, want_stats=True, bn=b.bn2.desc(t.st2, N * hw, self.device))
            else:
                _, st = b.conv2.forward(t.a1, t.r2, want_stats=True)
                b.bn2.finalize(st, N * hw, t.st2, train)
            ops.bnact_fwd(t.r2, t.out, scale=t.st2.scale, shift=t.st2.shift, residual=h)
            h = t.out
        if train:
            self.mid.forward(h, c.rm, want_stats=True, bn=self.mid_bn.desc(c.st_m, N * hw, self.device))
        else:
            _, st = self.mid.forward(h, c.rm, want_stats=True)
            self.mid_bn.finalize(st, N * hw, c.st_m, train)
        ops.bnact_fwd(c.rm, c.hm, scale=c.st_m.scale, shift=c.st_m.shift, residual=c.h0)
        h = c.hm
        for j, (conv, slope) in enumerate(self.sub):
            conv.forward(h, c.s_raw[j])
            ops.prelu_fwd(c.s_raw[j], slope.data, c.s_act[j], shuffle=2)
            h = c.s_act[j]
        self.last.forward(h, c.out, act=ACT_TANH)
        return c

    def backward(self, c, g_feat=None, wgrad=True):
        """c.g_out holds dL/d(image); g_feat: optional gradients w.r.t. features(c)"""
        G = self._gbufs(c.N, c.H, c.W)
        gf = dict(zip(self.hook_blocks, g_feat)) if g_feat is not None else {}
        ds = lambda p: p.grad if (wgrad and self.train_prelu) else None
        ops.bnact_bwd(c.out, None, c.g_out, c.g_out, in_act=ACT_TANH)
        wg = WgradCollector(self, c)       # the 3 x 3 / 9 x 9 layers' weight gradients as grouped launches: tail + mid, then the trunk
        if wgrad:
            wg.add(self.last, c.s_act[1], c.g_out)
        self.last.backward_data(c.g_out, G.s_act[1])
        for j in (1, 0):
            conv, slope = self.sub[j]
            ops.prelu_bwd(c.s_raw[j], slope.data, G.s_act[j], G.s_raw[j], dslope=ds(slope), shuffle=2)
            src = c.s_act[0] if j == 1 else c.hm
            if wgrad:
                wg.add(conv, src, G.s_raw[j])
            conv.backward_data(G.s_raw[j], G.s_act[0] if j == 1 else G.hm)
        # hm = BN(mid(h_last)) + h0
        bn = self.mid_bn.bn
        ops.nhwc_copy(G.hm, 0, G.h0, 0, self.C)                 # long skip: dL/dh0 starts with dL/dhm
        ops.bnact_bwd(c.rm, None, G.hm, G.rm, bn=c.st_m, gamma=bn.weight.data, beta=bn.bias.data, bn_eval=not c.train,
                      dgamma=bn.weight.grad if wgrad else None, dbeta=bn.bias.grad if wgrad else None)
        h_last = c.blk[-1].out if self.blocks else c.h0
        if wgrad:
            wg.add(self.mid, h_last, G.rm)
        self.mid.backward_data(G.rm, G.h)
        wg.flush('tail')
        for bi in range(len(self.blocks) - 1, -1, -1):
            b, t = self.blocks[bi], c.blk[bi]
            g_r2, g_a1, g_z1, g_r1 = G.blk[bi]
            h_in = c.blk[bi - 1].out if bi > 0 else c.h0
            if bi in gf and gf[bi] is not None:
                ops.nhwc_add(gf[bi], 0, G.h, 0, self.C)
            bn2, bn1 = b.bn2.bn, b.bn1.bn
            ops.bnact_bwd(t.r2, None, G.h, g_r2, bn=t.st2, gamma=bn2.weight.data, beta=bn2.bias.data, bn_eval=not c.train,
                          dgamma=bn2.weight.grad if wgrad else None, dbeta=bn2.bias.grad if wgrad else None)
            if wgrad:
                wg.add(b.conv2, t.a1, g_r2)
            b.conv2.backward_data(g_r2, g_a1)
            ops.prelu_bwd(t.z1, b.slope.data, g_a1, g_z1, dslope=ds(b.slope))
            ops.bnact_bwd(t.r1, None, g_z1, g_r1, bn=t.st1, gamma=bn1.weight.data, beta=bn1.bias.data, bn_eval=not c.train,
                          dgamma=bn1.weight.grad if wgrad else None, dbeta=bn1.bias.grad if wgrad else None)
            if wgrad:
                wg.add(b.conv1, h_in, g_r1)
            b.conv1.backward_data(g_r1, G.tmp)
            ops.nhwc_add(G.tmp, 0, G.h, 0, self.C)
            # a group per four blocks: its launch runs beside the next blocks' chain, as the per-layer launches did (one group for
            # the whole trunk runs behind the pass: +1.4 ms on the 96 -> 384 step, profiles/r6_wgrad_group.txt)
            if (len(self.blocks) - bi) % 4 == 0:
                wg.flush('trunk%d' % bi)
        ops.nhwc_add(G.h, 0, G.h0, 0, self.C)
        ops.prelu_bwd(c.raw0, self.first_slope.data, G.h0, G.raw0, dslope=ds(self.first_slope))
        if wgrad:
            wg.add(self.first, c.x_in, G.raw0)
        wg.flush('trunk')
        ops.SideStream.get(self.device).join()


class SRDiscriminatorEngine:
    """Discriminator / MaskDiscriminator (models/SRGAN.py:201-297): conv3 blocks (stride 1, 2, 1, 2 ...) with BatchNorm
    from the second on, [gate], LeakyReLU(0.2); global average pool; Linear(C, 1)"""

    def __init__(self, module, masked, threshold, device):
        self.module, self.masked, self.tau, self.device = module, masked, float(threshold), device
        self.conv, self.bn, self.gate = [], [], []
        for i, blk in enumerate(module.conv_blocks.children()):
            layers = list(blk.conv_block.children())
            self.conv.append(ConvOp(layers[0].weight, layers[0].bias, 3, 1 if i % 2 == 0 else 2, 1, False))
            self.bn.append(BNOp(layers[1]) if isinstance(layers[1], nn.BatchNorm2d) else None)
            gates = [m for m in layers if hasattr(m, 'alpha')]
            self.gate.append(gates[0] if gates else None)
        self.fc = module.fc1
        self.L = len(self.conv)
        self.chan = [c.rows for c in self.conv]
        self.mask = [torch.ones(w, dtype=torch.float32, device=device) if g is not None else None
                     for w, g in zip(self.chan, self.gate)]
        self.hook_layers = (1, 3)
        self.ctx, self.gbuf = {}, {}

    def convs(self):
        return self.conv

    def repack(self):
        if getattr(self, '_pack', None) is None:
            self._pack = ops.PackPlan(self.conv, self.device)
        self._pack.run()

    def refresh_masks(self):
        refresh_gate_masks(self)

    def _sizes(self, H, W):
        out = []
        for i in range(self.L):
            if i % 2 == 1:
                H, W = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            out.append((H, W))
        return out

    def new_ctx(self, N, H, W, tag):
        key = (N, H, W, tag)
        if key in self.ctx:
            return self.ctx[key]
        dev = self.device
        c = type('SRDCtx', (), {})()
        c.N, c.H, c.W = N, H, W
        c.hs = self._sizes(H, W)
        c.x_in = ops.new_act(N, 3, H, W, dev)
        c.raw = [ops.new_act(N, self.chan[i], *c.hs[i], dev) for i in range(self.L)]
        c.act = [ops.new_act(N, self.chan[i], *c.hs[i], dev) for i in range(self.L)]
        c.st = [ops.BNState(self.chan[i], dev) if self.bn[i] is not None else None for i in range(self.L)]
        c.pooled = torch.zeros((N, self.chan[-1]), dtype=torch.float32, device=dev)
        c.pred = ops.new_act(N, 1, 1, 1, dev)
        c.train = True
        self.ctx[key] = c
        return c

    def _gbufs(self, N, H, W):
        key = (N, H, W)
        if key not in self.gbuf:
            dev = self.device
            hs = self._sizes(H, W)
            g = type('SRDGrad', (), {})()
            g.pred = ops.new_act(N, 1, 1, 1, dev)
            g.act = [ops.new_act(N, self.chan[i], *hs[i], dev) for i in range(self.L)]
            g.raw = [ops.new_act(N, self.chan[i], *hs[i], dev) for i in range(self.L)]
            g.x_in = ops.new_act(N, 3, H, W, dev)
            self.gbuf[key] = g
        return self.gbuf[key]

    def features(self, c):
        return [c.act[i] for i in self.hook_layers]

    def grad_pred_buffer(self, c):
        return self._gbufs(c.N, c.H, c.W).pred

    def forward(self, c, train=True):
        c.train = train
        if self.masked:
            self.refresh_masks()
        src = c.x_in
        for i in range(self.L):
            if self.bn[i] is not None:
                n = c.N * c.hs[i][0] * c.hs[i][1]
                if train:
                    self.conv[i].forward(src, c.raw[i], want_stats=True, bn=self.bn[i].desc(c.st[i], n, self.device))
                else:
                    _, st = self.conv[i].forward(src, c.raw[i], want_stats=True)
                    self.bn[i].finalize(st, n, c.st[i], train)
                ops.bnact_fwd(c.raw[i], c.act[i], scale=c.st[i].scale, shift=c.st[i].shift, gate=self.mask[i], act=ACT_LRELU)
            else:
                self.conv[i].forward(src, c.raw[i])
                ops.bnact_fwd(c.raw[i], c.act[i], gate=self.mask[i], act=ACT_LRELU)
            src = c.act[i]
        ops.pool_linear_fwd(src, self.fc.weight.data, self.fc.bias.data, c.pooled, c.pred)
        return c.pred

    def backward(self, c, has_pred_grad=True, g_feat=None, wgrad=True, agrad=False, need_dx=True):
        G = self._gbufs(c.N, c.H, c.W)
        g_feat = g_feat or [None, None]
        feat_of = dict(zip(self.hook_layers, g_feat))
        L = self.L
        if has_pred_grad:
            ops.pool_linear_bwd(G.pred, self.fc.weight.data, c.pooled, c.act[L - 1], dx=G.act[L - 1],
                                dw=self.fc.weight.grad if wgrad else None, db=self.fc.bias.grad if wgrad else None)
        for i in range(L - 1, -1, -1):
            gf = feat_of.get(i)
            if i == L - 1 and not has_pred_grad:
                assert gf is not None, 'nothing to back-propagate'
                ops.nhwc_copy(gf, 0, G.act[i], 0, self.chan[i])
            elif gf is not None:
                ops.nhwc_add(gf, 0, G.act[i], 0, self.chan[i])
            gate = self.gate[i]
            da = gate.alpha.grad if (agrad and gate is not None) else None
            if self.bn[i] is not None:
                bn = self.bn[i].bn
                # (y = None: the activation's sign is recomputed from the BatchNorm affine instead of reading the saved output -- one
                # tensor less to read in each of the two passes; these are 38-151 MB at the 96 -> 384 size)
                ops.bnact_bwd(c.raw[i], None, G.act[i], G.raw[i], bn=c.st[i], gamma=bn.weight.data, beta=bn.bias.data,
                              bn_eval=not c.train, gate=self.mask[i], act=ACT_LRELU, dgamma=bn.weight.grad if wgrad else None,
                              dbeta=bn.bias.grad if wgrad else None, dalpha=da)
            else:
                ops.bnact_bwd(c.raw[i], None if self.mask[i] is not None else c.act[i], G.act[i], G.raw[i], gate=self.mask[i], act=ACT_LRELU,
                              dalpha=da)              # (gated: the activation's sign from raw x gate, the 302 MB saved output is not read)
            src = c.x_in if i == 0 else c.act[i - 1]
            if wgrad:
                self.conv[i].backward_weight(src, G.raw[i])
            if i > 0:
                self.conv[i].backward_data(G.raw[i], G.act[i - 1])
            elif need_dx:
                self.conv[0].backward_data(G.raw[0], G.x_in)
        ops.SideStream.get(self.device).join()
        return G.x_in if need_dx else None


class VGGEngine:
    """TruncatedVGG19 (models/GANLoss.py:95-145): frozen conv3 + ReLU stack with 2x2 max pools; forward and the data
    gradient only (the weights never train)."""

    def __init__(self, module, device):
        self.device = device
        self.layers = []                      # ('conv', ConvOp) | ('pool', None); ReLU is fused into the conv
        for m in module.truncated_vgg19.children():
            if isinstance(m, nn.Conv2d):
                self.layers.append(('conv', ConvOp(m.weight, m.bias, 3, 1, 1, False)))
            elif isinstance(m, nn.MaxPool2d):
                self.layers.append(('pool', None))
        self.ctx, self.gbuf = {}, {}

    def repack(self):
        ops.PackPlan([c for k, c in self.layers if k == 'conv'], self.device).run()

    def _shapes(self, H, W):
        out, C = [], 3
        for kind, conv in self.layers:
            if kind == 'conv':
                C = conv.rows
            else:
                H, W = H // 2, W // 2
            out.append((C, H, W))
        return out

    def new_ctx(self, N, H, W, tag):
        key = (N, H, W, tag)
        if key not in self.ctx:
            c = type('VGGCtx', (), {})()
            c.N, c.H, c.W = N, H, W
            c.x_in = ops.new_act(N, 3, H, W, self.device)
            c.act = [ops.new_act(N, C, h, w, self.device) for C, h, w in self._shapes(H, W)]
            self.ctx[key] = c
        return self.ctx[key]

    def forward(self, c):
        src = c.x_in
        for (kind, conv), out in zip(self.layers, c.act):
            if kind == 'conv':
                conv.forward(src, out, act=ACT_RELU)
            else:
                ops.maxpool_fwd(src, out)
            src = out
        return src

    def backward(self, c, g_last):
        """g_last: dL/d(output feature map), consumed in place; returns dL/d(x_in)"""
        key = (c.N, c.H, c.W)
        if key not in self.gbuf:
            self.gbuf[key] = ([ops.new_act(c.N, C, h, w, self.device) for C, h, w in self._shapes(c.H, c.W)],
                              ops.new_act(c.N, 3, c.H, c.W, self.device))
        gact, gx = self.gbuf[key]
        g = g_last
        for i in range(len(self.layers) - 1, -1, -1):
            kind, conv = self.layers[i]
            src = c.act[i - 1] if i > 0 else c.x_in
            dst = gact[i - 1] if i > 0 else gx
            if kind == 'conv':
                # (behind a max pool the ReLU's backward was applied by the pool's: maxpool_bwd(relu_mask=True) below)
                if not (i + 1 < len(self.layers) and self.layers[i + 1][0] == 'pool'):
                    ops.bnact_bwd(c.act[i], None, g, g, in_act=ACT_RELU)
                conv.backward_data(g, dst)
            else:
                # src = the ReLU'd output of the conv in front of the pool (always a conv in VGG19): its mask rides along
                ops.maxpool_bwd(src, g, dst, relu_mask=i > 0 and self.layers[i - 1][0] == 'conv')
            g = dst
        return gx
