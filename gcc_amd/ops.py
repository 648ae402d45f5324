"""Thin Python wrappers over the C ABI (include/gcc_hip.h) on torch device tensors.

Activations are ``torch.bfloat16`` tensors of logical shape [N, C, H, W] in channels_last memory
(NHWC), possibly channel-slices of a wider buffer; PyTorch only provides the memory, the current
HIP stream and (elsewhere) torch.distributed -- every computation below is a libgcc_hip.so kernel.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, check  # noqa: F401

_ws_cache = {}


class _Profile:
    """Optional per-launch timing of the MFMA kernels with HIP events on the launch stream
    (bench.py's roofline block).  Inactive by default: zero overhead on the product path."""

    def __init__(self):
        self.active = False
        self.spans_only = False         # time engine passes (span) of the production path; no launch is bracketed, no route changes
        self.records = []
        self.streaming = os.environ.get('GCC_PROFILE_STREAMING') == '1'     # also bracket the BN backward launches

    def start(self, steps=None):
        """steps: stop bracketing launches after this many step_done() calls (keeps the event
        overhead -- about 6% when every launch of a step is bracketed -- out of most of the timed region)"""
        self.records = []
        self.spans = []
        self.tag = None
        self.active = True
        self.steps_left = steps
        self.steps_seen = 0

    def start_spans(self):
        """spans alone: the passes run exactly the launches the product path runs (`active` stays False, so no fused route is
        replaced by its bracketable parts); stop_spans() returns {name: seconds}"""
        self.spans = []
        self.tag = None
        self.spans_only = True

    def stop_spans(self):
        self.spans_only = False
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.spans:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) * 1e-3
        self.spans = []
        return out

    def step_done(self):
        if self.active:
            self.steps_seen += 1
            if self.steps_left is not None:
                self.steps_left -= 1
                if self.steps_left <= 0:
                    self.active = False

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, kind, flops, e0, shape=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((kind, flops, e0, e1, shape, self.tag))

    def span(self, name):
        """context manager: wall duration of a whole engine pass (every kernel in it, MFMA or not) under `name`; conv
        launches inside are attributed to `name` too (PROFILE.tag)"""
        return _Span(self, name)

    def stop(self, peak=2.5e15, hbm=8.0e12):
        self.active = False
        torch.cuda.synchronize()
        agg, shapes = {}, {}
        tags = {}
        alg_bytes = {}
        roof_s, roof_meas_s, roof_n = 0.0, 0.0, 0
        for kind, flops, e0, e1, shape, tag in self.records:
            sec = e0.elapsed_time(e1) * 1e-3
            if shape is not None and len(shape) == 8 and shape[0] in ('fprop', 'dgrad', 'wgrad'):
                # SURVEY.md 8(d) 'conv roofline': per layer max(FLOP / MFMA peak, minimum bytes / HBM bandwidth), summed --
                # minimum bytes = every operand once (bf16 activations and weights; the fp32 weight gradient of a wgrad)
                op, n_, ho, wo, ci, co, k_, st_ = shape
                small, big = n_ * ho * wo * co, n_ * (ho * st_) * (wo * st_) * ci      # conv output side / input side
                wts = co * ci * k_ * k_
                nbytes = 2.0 * (small + big) + (4.0 if op == 'wgrad' else 2.0) * wts
                alg_bytes[kind] = alg_bytes.get(kind, 0.0) + nbytes
                roof_s += max(flops / peak, nbytes / hbm)
                roof_meas_s += sec
                roof_n += 1
            if tag is not None:
                t = tags.setdefault(tag, [0.0, 0.0, 0])
                t[0] += flops
                t[1] += sec
                t[2] += 1
            a = agg.setdefault(kind, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += sec
            a[2] += 1
            b = shapes.setdefault((kind.split(' ')[0], shape, tag or ''), [0.0, 0.0, 0])
            b[0] += flops
            b[1] += sec
            b[2] += 1
        self.records = []
        self.tag_stats = {k: {'flop': v[0], 'conv_s': v[1], 'conv_launches': v[2]} for k, v in tags.items()}
        for name, e0, e1 in getattr(self, 'spans', []):
            t = self.tag_stats.setdefault(name, {'flop': 0.0, 'conv_s': 0.0, 'conv_launches': 0})
            t['span_s'] = t.get('span_s', 0.0) + e0.elapsed_time(e1) * 1e-3
            t['spans'] = t.get('spans', 0) + 1
        self.spans = []
        if os.environ.get('GCC_PROFILE_SHAPES') == '1':      # per-geometry table on stderr (tuning aid)
            import sys
            for (kind, shape, tag), (fl, sec, n) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('GCC_PROFILE_TOP', '200'))]:
                print('%-14s %-44s %-14s n=%4d  total %7.3f ms  avg %7.1f us  %7.1f TFLOP/s' % (
                    kind, shape, tag, n, sec * 1e3, sec / n * 1e6, fl / sec / 1e12), file=sys.stderr)
        if not agg:
            return None
        dom = max(agg, key=lambda k: agg[k][1])
        out = None
        per = {}
        for kind, (fl, sec, n) in agg.items():
            per[kind] = {'launches': n, 'avg_us': round(1e6 * sec / n, 2), 'tflops': round(fl / sec / 1e12, 2),
                         'time_s': round(sec, 4), 'gflop': round(fl / 1e9, 3)}
            if kind == dom:
                out = {'bound': 'mfma', 'kernel': kind, 'achieved': round(fl / sec / 1e12, 2), 'peak': peak / 1e12,
                       'unit': 'TFLOP/s', 'frac': round(fl / sec / peak, 4), 'traffic': None, 'launches': n,
                       'steps_bracketed': getattr(self, 'steps_seen', None),
                       'avg_launch_us': round(1e6 * sec / n, 2),
                       'algorithmic_gflop_per_launch': round(fl / n / 1e9, 3),
                       # every operand once (bf16 input + output + weights of the launch): what `traffic` is to be read against
                       'algorithmic_bytes_per_launch': round(alg_bytes.get(kind, 0.0) / n)}
        out['per_kernel'] = per
        if roof_n:
            out['conv_roofline'] = {'layers': roof_n, 'bound_ms': round(roof_s * 1e3, 3), 'measured_ms': round(roof_meas_s * 1e3, 3),
                                    'frac': round(roof_s / roof_meas_s, 4),
                                    'definition': 'sum over the conv launches of one iteration of max(FLOP / 2.5 PFLOP/s, '
                                                  'minimum operand bytes / 8 TB/s) over the sum of their measured durations'}
        return out


class _Span:
    def __init__(self, prof, name):
        self.prof, self.name = prof, name

    def __enter__(self):
        if self.prof.active or self.prof.spans_only:
            self.prev = self.prof.tag
            self.prof.tag = self.name
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if getattr(self, 'e0', None) is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.prof.spans.append((self.name, self.e0, e1))
            self.prof.tag = self.prev
        return False


PROFILE = _Profile()
# gcc_conv_route() -> label of the bracketed launch
_ROUTE_KIND = {0: 'igemm_kernel (conv fprop/dgrad)', 1: 'thin_fprop / thin_dgrad (image layers)', 2: 'head route (igemm 128x16 + tap sum)',
               3: 'thinout_fprop / thinout_dgrad (wide filters, <= 3 output channels)',
               4: 'ring3_kernel (3 x 3 stride 1, <= 64 channels both sides)', -1: 'igemm_kernel (conv fprop/dgrad)'}


def lib():
    return _lib.load()


_dev_index = None


def set_device_index(idx):
    """the device whose current stream ops are enqueued on (one GPU per process; gdist.local_device sets it)"""
    global _dev_index
    _dev_index = idx


# torch's private fast paths for "which stream is current" / "make this stream current" (torch 2.10: _cuda_getCurrentRawStream,
# _cuda_setStream), each behind a guard: if a torch upgrade removes or re-signs one, the public API takes over (torch.cuda.
# current_stream().cuda_stream, torch.cuda.set_stream) -- slower on the host (~8 / ~20 us per call), never differently ordered.
# tests/test_kernels_gpu.py::test_stream_helpers_order_work asserts the ordering through whichever path is active.
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_SET_STREAM = getattr(torch._C, '_cuda_setStream', None)
if os.environ.get('GCC_PUBLIC_STREAM_API') == '1':          # test hook: exercise the fallbacks
    _RAW_STREAM = _SET_STREAM = None


def _set_current(s):
    global _SET_STREAM
    if _SET_STREAM is not None:
        try:
            _SET_STREAM(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
            return
        except (TypeError, AttributeError):
            _SET_STREAM = None
    torch.cuda.set_stream(s)


def stream():
    """raw hipStream_t of the current stream (the direct C call: this runs once per launch on the host's hot path)"""
    global _dev_index, _RAW_STREAM
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    if _RAW_STREAM is not None:
        try:
            return _RAW_STREAM(_dev_index)
        except (TypeError, AttributeError):
            _RAW_STREAM = None
    return torch.cuda.current_stream(_dev_index).cuda_stream


_stream_objs = {}


def current_stream():
    """torch.cuda.Stream object of the current stream.  torch.cuda.current_stream() takes ~8 us of host time (device-index
    resolution, availability checks); this is a dict lookup on the raw handle -- the hot path asks a few hundred times per
    iteration (event records / waits, side-stream forks)."""
    raw = stream()
    s = _stream_objs.get(raw)
    if s is None:
        s = _stream_objs[raw] = torch.cuda.current_stream()
    return s


class on_stream:
    """`with on_stream(s)`: torch.cuda.stream(s) for the one device of this process, without its per-entry device and
    current-stream queries (~20 us of host time per block)"""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        s = self.s
        if s:
            self.prev = current_stream()
            _set_current(s)
        return s

    def __exit__(self, *exc):
        if self.s:
            _set_current(self.prev)
        return False


def ceil8(v):
    return (v + 7) & ~7


def workspace(nbytes, device, slot='default'):
    """Grow-only scratch buffer per (device, slot, stream): all users of one buffer are ordered on one stream (the online
    teacher may run on its own stream next to the student's)."""
    key = (device, slot, stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


class Event:
    """An event of the library (gcc_event_*, hipEventDisableTiming): records and waits made through it are part of a launch
    recording (gcc_amd.replay).  Never destroyed: a recording may hold its handle; a process creates a few dozen.
    record / wait take a torch.cuda.Stream (default: the current stream); the *_raw forms take the hipStream_t itself (a
    side-stream fork is two of these calls per weight-gradient launch: the property lookups count)."""
    __slots__ = ('h',)

    def __init__(self):
        h = C.c_void_p()
        check(lib().gcc_event_create(C.byref(h)), 'gcc_event_create')
        self.h = h.value

    def record(self, s=None):
        if _lib.load().gcc_event_record(self.h, s.cuda_stream if s is not None else stream()):
            raise RuntimeError('gcc_event_record failed')

    def wait(self, s=None):
        if _lib.load().gcc_stream_wait_event(s.cuda_stream if s is not None else stream(), self.h):
            raise RuntimeError('gcc_stream_wait_event failed')

    def record_raw(self, raw):
        if _lib.load().gcc_event_record(self.h, raw):
            raise RuntimeError('gcc_event_record failed')

    def wait_raw(self, raw):
        if _lib.load().gcc_stream_wait_event(raw, self.h):
            raise RuntimeError('gcc_stream_wait_event failed')


def wait_event(s, ev):
    """stream s waits for ev: an ops.Event, or a torch.cuda.Event handed in from outside (a data loader's 'ready')"""
    if isinstance(ev, Event):
        ev.wait(s)
    else:
        s.wait_event(ev)


_pair_events = {}


def wait_stream(s, other):
    """stream s waits for everything enqueued on `other` so far (torch's Stream.wait_stream through the library's events)"""
    key = (s.cuda_stream, other.cuda_stream)
    ev = _pair_events.get(key)
    if ev is None:
        ev = _pair_events[key] = Event()
    ev.record(other)
    ev.wait(s)


def note_host(obj):
    """host-side bookkeeping of the iteration that no launch carries (BatchNorm's num_batches_tracked count): while
    recording, obj.replay_update(recording, 0) is asked to repeat it for every replayed iteration"""
    if RECORDING:
        _dynamic.append((obj, 0))


# True while gcc_amd.replay records an iteration: host-side shortcuts that would hide work from the recording thread (the
# teacher's enqueue thread) are off, and launches that carry per-iteration scalars announce themselves (note_dynamic)
RECORDING = False
_dynamic = []          # while recording: (object, tag) in launch order; object.replay_update(recording, tag) runs before every replay
_tag_counter = 0


def note_dynamic(obj):
    """call right before a launch whose BY-VALUE arguments change from iteration to iteration (Adam's bias corrections, the
    image pool's draws): while recording, the launch is tagged and obj.replay_update(recording, tag) is asked to patch it
    before every replay.  No-op otherwise."""
    global _tag_counter
    if RECORDING:
        _tag_counter += 1
        lib().gcc_replay_tag_next(_tag_counter)
        _dynamic.append((obj, _tag_counter))


class SideStream:
    """A second HIP stream per device: MFMA-bound weight-gradient kernels run on it while the
    HBM-bound BatchNorm / activation backward chain continues on the main stream."""
    _inst = {}

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.raw = self.stream.cuda_stream
        self.dirty = False
        self.ev_fork, self.ev_join = Event(), Event()     # re-recorded: a wait holds the record it saw

    @classmethod
    def get(cls, device):
        """the side stream that belongs to the CURRENT stream of `device`"""
        key = (device, stream())
        inst = cls._inst.get(key)
        if inst is None:
            inst = cls._inst[key] = SideStream(device)
        return inst

    @classmethod
    def share(cls, device, user, owner):
        """the stream `user` (raw handle) gets a SideStream object that launches on the SAME HIP stream as `owner`'s (own events,
        own dirty flag): two chains that run side by side and accumulate into the same gradient buffers keep one launch order
        for their weight-gradient kernels -- the host's enqueue order -- instead of racing on two side streams; and no further
        HIP stream is created (the device has four hardware queues)"""
        cur = cls._inst.get((device, user))
        own = cls._inst.get((device, owner))
        if own is None:
            own = cls._inst[(device, owner)] = SideStream(device)
        if cur is None or cur.raw != own.raw:
            inst = cls.__new__(cls)
            inst.stream, inst.raw, inst.dirty = own.stream, own.raw, False
            inst.ev_fork, inst.ev_join = Event(), Event()
            cls._inst[(device, user)] = inst
        return cls._inst[(device, user)]

    def fork(self):
        """everything enqueued on the main stream so far happens-before later side-stream work"""
        self.ev_fork.record_raw(stream())
        self.ev_fork.wait_raw(self.raw)
        self.dirty = True

    def join(self):
        """the main stream waits for all side-stream work enqueued so far"""
        if self.dirty:
            self.ev_join.record_raw(self.raw)
            self.ev_join.wait_raw(stream())
            self.dirty = False


def new_act(N, Cc, H, W, device, ld=None):
    """Zero-initialised NHWC bf16 activation [N, Cc, H, W] with pixel stride ld >= ceil8(Cc)."""
    ld = ld or ceil8(Cc)
    base = torch.zeros((N, H, W, ld), dtype=torch.bfloat16, device=device)
    return base.permute(0, 3, 1, 2)[:, :Cc]


def cslice(t, off, Cc):
    """Channel slice [off, off+Cc) of an NHWC activation (off % 8 == 0)."""
    assert off % 8 == 0
    return t[:, off:off + Cc]


def geom(t):
    """(ptr, N, C, H, W, ld) of an NHWC bf16 activation view.  The iteration's activations are persistent buffers: the tuple is
    kept on the tensor object (a view made per call is simply computed again; geometry and address of a tensor object never
    change -- nothing in this package resizes or re-homes an activation in place)."""
    d = t.__dict__
    g = d.get('_gcc_geom')
    if g is not None:
        if GEOM_CHECK and g != _geom(t):       # GCC_DEBUG_GEOM=1 (tests/conftest.py sets it): the invariant above, enforced
            raise AssertionError('activation re-homed or resized in place: cached %r, now %r' % (g, _geom(t)))
        return g
    g = d['_gcc_geom'] = _geom(t)
    return g


GEOM_CHECK = os.environ.get('GCC_DEBUG_GEOM', '0') != '0'


def _geom(t):
    N, Cc, H, W = t.shape
    s0, s1, s2, ld = t.stride()
    p = t.data_ptr()
    if (t.dtype is not torch.bfloat16 or (s1 != 1 and Cc != 1) or (H > 1 and s2 != W * ld) or (N > 1 and s0 != H * W * ld)
            or (ld & 7) or (p & 15)):
        raise AssertionError('not an NHWC bf16 activation view: %s %s %s' % (t.dtype, tuple(t.shape), t.stride()))
    return p, N, Cc, H, W, ld


# ---- the tile plan of the convolution calls (include/gcc_hip.h gcc_conv_plan_t) ------------------------------------------
# It travels with every call (conv_desc) instead of living in the library as process-wide options (rounds 2-4).  A model class
# states the plan of its schedule at the head of each phase (models/_streams.py _ensure_plan); the statement is THREAD-local, so
# the teacher's enqueue thread and the main thread -- or two models of different schedules -- never see each other's.
# GCC_IGEMM_TILES / GCC_IGEMM_BIG_MIN / GCC_IGEMM_BIG_NK / GCC_IGEMM_PAIR / GCC_HALO_HC / GCC_WGRAD_WGS_BIG / GCC_WGRAD_WGS in
# the environment pin a field for A/B runs (an explicit value wins over what the schedule asks for, as before).
_PLAN_ENV = {'tile_families': 'GCC_IGEMM_TILES', 'big_min': 'GCC_IGEMM_BIG_MIN', 'big_nk': 'GCC_IGEMM_BIG_NK', 'pair': 'GCC_IGEMM_PAIR',
             'halo_hc': 'GCC_HALO_HC', 'wgrad_wgs_big': 'GCC_WGRAD_WGS_BIG', 'wgrad_wgs': 'GCC_WGRAD_WGS'}
_plan_pinned = {f: int(os.environ[e]) for f, e in _PLAN_ENV.items() if os.environ.get(e, '') not in ('', '-1')}
_plan_tls = threading.local()
_PLAN_DEFAULT = (0,) * len(_lib.PLAN_FIELDS)


def set_plan(**fields):
    """state the calling thread's tile plan: named fields of gcc_conv_plan_t, everything else back to the library's default (0).
    Returns the previous plan (a tuple for restore_plan)."""
    prev = getattr(_plan_tls, 'plan', _PLAN_DEFAULT)
    unknown = set(fields) - set(_lib.PLAN_FIELDS)
    assert not unknown, unknown
    fields.update(_plan_pinned)
    _plan_tls.plan = tuple(int(fields.get(f, 0)) for f in _lib.PLAN_FIELDS)
    return prev


def restore_plan(prev):
    _plan_tls.plan = prev


def current_plan():
    """{field: value} of the calling thread's plan"""
    plan = getattr(_plan_tls, 'plan', None)
    if plan is None:
        set_plan()
        plan = _plan_tls.plan
    return dict(zip(_lib.PLAN_FIELDS, plan))


class plan_override:
    """with plan_override(halo_hc=1): ... -- the current plan with some fields changed, for the calls inside (environment pins win)"""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        cur = current_plan()
        cur.update(self.fields)
        self.prev = set_plan(**cur)

    def __exit__(self, *exc):
        restore_plan(self.prev)
        return False


def conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy):
    plan = getattr(_plan_tls, 'plan', None)
    if plan is None:
        set_plan()
        plan = _plan_tls.plan
    return _lib.conv_t(N, H, W, Ci, Co, k, k, stride, pad, ldx, 0, ldy, 0, plan)


def pack_weights(master, want_w=True, want_wt=True):
    """master: fp32 [rows, cols, KH, KW] channels_last parameter (physical [rows][taps][cols])."""
    rows, cols, KH, KW = master.shape
    m = master.detach()
    assert m.dtype == torch.float32
    if not m.is_contiguous(memory_format=torch.channels_last) and KH * KW > 1:
        raise _lib.GccError('conv master weights must be channels_last')
    taps = KH * KW
    w = torch.empty((rows, taps, ceil8(cols)), dtype=torch.bfloat16, device=m.device) if want_w else None
    wt = torch.empty((cols, taps, ceil8(rows)), dtype=torch.bfloat16, device=m.device) if want_wt else None
    check(lib().gcc_pack_weights(m.data_ptr(), rows, taps, cols, w.data_ptr() if want_w else None,
                                 wt.data_ptr() if want_wt else None, stream()), 'gcc_pack_weights')
    return w, wt


def pack_weights_into(master, w, wt):
    rows, cols, KH, KW = master.shape
    check(lib().gcc_pack_weights(master.data_ptr(), rows, KH * KW, cols, w.data_ptr() if w is not None else None,
                                 wt.data_ptr() if wt is not None else None, stream()), 'gcc_pack_weights')


FUSED_PACK = os.environ.get('GCC_FUSED_PACK', '1') != '0'


class PackPlan:
    """One launch that refreshes the bf16 W / Wt packings of a list of convs (engine.ConvOp) from their
    fp32 masters.  Pointers must stay valid (flat parameter storage, persistent packings)."""

    def __init__(self, convs, device):
        convs = list(convs)
        D = (_lib.pack_desc_t * len(convs))()
        items = []
        for i, c in enumerate(convs):
            rows, cols, taps = c.rows, c.cols, c.k * c.k
            colsp, rowsp = c.cols_p, c.rows_p
            D[i] = _lib.pack_desc_t(c.weight.data_ptr(), c.w.data_ptr(), c.wt.data_ptr(), rows, taps, cols, colsp, rowsp,
                                    c.row_split, c.col_split, 0)
            if (FUSED_PACK and not c.row_split and not c.col_split and cols % 4 == 0 and c.weight.data_ptr() % 16 == 0
                    and c.w is not None and c.wt is not None):
                for tap in range(taps):             # kind 2: both packings of a 64 x 64 tile from one read of the master
                    for rb in range((rowsp + 63) // 64):
                        for cb in range((colsp + 63) // 64):
                            items.append((i, 2, tap, rb, cb))
                continue
            for a in range((rowsp * taps * colsp + 2047) // 2048):
                items.append((i, 0, a, 0, 0))
            for tap in range(taps):
                for rb in range((rowsp + 31) // 32):
                    for cb in range((colsp + 31) // 32):
                        items.append((i, 1, tap, rb, cb))
        import numpy as np
        arr = np.zeros((len(items), 6), dtype=np.int32)
        arr[:, :5] = np.asarray(items, dtype=np.int32).reshape(-1, 5)
        self.n = len(items)
        self.d_desc = torch.frombuffer(bytearray(bytes(D)), dtype=torch.uint8).to(device)
        self.d_items = torch.from_numpy(arr).to(device)
        self.ptrs = [c.weight.data_ptr() for c in convs]
        self.convs = convs

    def run(self):
        if [c.weight.data_ptr() for c in self.convs] != self.ptrs:
            raise _lib.GccError('parameter storage moved after the pack plan was built')
        check(lib().gcc_pack_weights_multi(self.d_desc.data_ptr(), self.d_items.data_ptr(), self.n, stream()),
              'gcc_pack_weights_multi')


def _epilogue(bias, act, slope, stats, d=None, dgrad=0, device=None, bn=None):
    wsp, wsb = None, 0
    if d is not None:
        need = lib().gcc_conv_workspace(C.byref(d), dgrad)
        if need:
            ws = workspace(need, device, 'splitk')
            wsp, wsb = ws.data_ptr(), ws.numel()
    return _lib.epilogue_t(bias.data_ptr() if bias is not None else None, act, slope,
                           stats.data_ptr() if stats is not None else None, wsp, wsb,
                           C.cast(C.pointer(bn), C.c_void_p) if bn is not None else None, None, 0, 0, 0, None)


Y2_RELU, Y2_GATE = 1, 2        # gcc_epilogue_t.y2_mode
CONV_Y2 = os.environ.get('GCC_CONV_Y2', '1') != '0'      # A/B hook: 0 = always the separate gcc_bnact_fwd launch


def conv_fprop(x, w, Co, k, stride, pad, out=None, bias=None, act=ACT_NONE, slope=0.2, want_stats=False, bn=None,
               y2=None, y2_mode=0, y2_gate=None):
    """bn: a gcc_bn_t (bn_desc) -- the BatchNorm behind this conv is finalized inside the call (needs want_stats).
    y2: a second output f(out) -- Y2_RELU: relu(out); Y2_GATE: out * y2_gate[c] -- written by the conv launch itself where the
    library can (the thin image-layer route), by a gcc_bnact_fwd launch behind it otherwise."""
    xp, N, Ci, H, W, ldx = geom(x)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = new_act(N, Co, Ho, Wo, x.device)
    yp, _, _, _, _, ldy = geom(out)
    d = conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy)
    stats = None
    if want_stats:
        tiles = lib().gcc_conv_stat_tiles(C.byref(d), 0)
        stats = torch.empty((tiles, 2, Co), dtype=torch.float32, device=x.device)
    ep = _epilogue(bias, act, slope, stats, d, 0, x.device, bn)
    y2_after = False
    if y2 is not None:
        y2p, _, _, _, _, ldy2 = geom(y2)
        ep.y2, ep.ldy2, ep.y2off, ep.y2_mode = y2p, ldy2, 0, y2_mode
        ep.y2_gate = y2_gate.data_ptr() if y2_gate is not None else None
        if not (CONV_Y2 and lib().gcc_conv_y2_supported(C.byref(d), 0, C.byref(ep))):
            ep.y2, y2_after = None, True
    e0 = PROFILE.begin() if PROFILE.active else None
    check(lib().gcc_conv_fprop(C.byref(d), xp, w.data_ptr(), yp, C.byref(ep), stream()), 'gcc_conv_fprop')
    if y2_after:
        if y2_mode == Y2_RELU:
            bnact_fwd(out, y2, act=ACT_RELU)
        else:
            bnact_fwd(out, y2, gate=y2_gate, gate_after_act=True)
    if e0 is not None:
        PROFILE.end(_ROUTE_KIND[lib().gcc_conv_route(C.byref(d), 0, C.byref(ep))], 2.0 * N * Ho * Wo * Co * k * k * Ci, e0,
                    shape=('fprop', N, Ho, Wo, Ci, Co, k, stride))
    return (out, stats) if want_stats else out


def conv_dgrad(dy, wt, Ci, H, W, k, stride, pad, out=None, bias=None, act=ACT_NONE, slope=0.2, want_stats=False, bn=None):
    """dx [N,Ci,H,W] = conv_backward_data(dy) == ConvTranspose2d forward."""
    yp, N, Co, Ho, Wo, ldy = geom(dy)
    assert Ho == (H + 2 * pad - k) // stride + 1 and Wo == (W + 2 * pad - k) // stride + 1
    if out is None:
        out = new_act(N, Ci, H, W, dy.device)
    xp, _, _, _, _, ldx = geom(out)
    d = conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy)
    stats = None
    if want_stats:
        tiles = lib().gcc_conv_stat_tiles(C.byref(d), 1)
        stats = torch.empty((tiles, 2, Ci), dtype=torch.float32, device=dy.device)
    ep = _epilogue(bias, act, slope, stats, d, 1, dy.device, bn)
    e0 = PROFILE.begin() if PROFILE.active else None
    check(lib().gcc_conv_dgrad(C.byref(d), yp, wt.data_ptr(), xp, C.byref(ep), stream()), 'gcc_conv_dgrad')
    if e0 is not None:
        PROFILE.end(_ROUTE_KIND[lib().gcc_conv_route(C.byref(d), 1, C.byref(ep))], 2.0 * N * Ho * Wo * Co * k * k * Ci, e0,
                    shape=('dgrad', N, Ho, Wo, Ci, Co, k, stride))
    return (out, stats) if want_stats else out


def conv_bn_act(dgrad, src, w, raw, k, stride, pad, bn_module, st, count, y, y2=None, act=ACT_NONE, act2=ACT_NONE, slope=0.2,
                drop_p=0.0, seed=0):
    """conv (dgrad=False: raw = conv(src); True: raw = conv_backward_data(src) == ConvTranspose2d forward) + BatchNorm with the
    batch statistics of raw (running statistics of `bn_module` updated, mean / rstd / scale / shift into the BNState `st`) +
    y = act(drop(bn(raw))) [+ y2 = act2(..)] in one C call (gcc_conv_bn_act)"""
    sp, N, Cs, Hs, Ws, lds = geom(src)
    rp, _, Cr, Hr, Wr, ldr = geom(raw)
    if not dgrad:
        d = conv_desc(N, Hs, Ws, Cs, Cr, k, stride, pad, lds, ldr)
        flops = 2.0 * N * Hr * Wr * Cr * k * k * Cs
    else:
        assert Hs == (Hr + 2 * pad - k) // stride + 1 and Ws == (Wr + 2 * pad - k) // stride + 1
        d = conv_desc(N, Hr, Wr, Cr, Cs, k, stride, pad, ldr, lds)
        flops = 2.0 * N * Hs * Ws * Cs * k * k * Cr
    need = lib().gcc_conv_bn_act_workspace(C.byref(d), int(dgrad))
    ws = workspace(need, src.device, 'convbn')
    bn = bn_desc(bn_module, st, count, src.device)
    yp, ldy = (None, 0)
    if y is not None:
        yp, _, _, _, _, ldy = geom(y)
    y2p, ldy2 = (None, 0)
    if y2 is not None:
        y2p, _, _, _, _, ldy2 = geom(y2)
    a = _lib.bnact_t(None, None, None, 0, act, slope, act2, drop_p, seed, 1, 0, None)
    e0 = PROFILE.begin() if PROFILE.active else None
    check(lib().gcc_conv_bn_act(C.byref(d), int(dgrad), sp, w.data_ptr(), rp, C.byref(bn), C.byref(a), yp, ldy, 0, y2p, ldy2, 0,
                                ws.data_ptr(), ws.numel(), stream()), 'gcc_conv_bn_act')
    if e0 is not None:
        PROFILE.end('conv + BatchNorm + activation (one call)', flops, e0,
                    shape=('dgrad+bn' if dgrad else 'fprop+bn', N, Hs if dgrad else Hr, Ws if dgrad else Wr, Cr if dgrad else Cs,
                           Cs if dgrad else Cr, k, stride))


def conv_wgrad(x, dy, dw, k, stride, pad, accumulate=False):
    """dw: fp32 [Co, Ci, k, k] channels_last gradient buffer (physical [Co][taps][Ci])."""
    xp, N, Ci, H, W, ldx = geom(x)
    yp, _, Co, Ho, Wo, ldy = geom(dy)
    assert tuple(dw.shape) == (Co, Ci, k, k) and dw.dtype == torch.float32
    assert k == 1 or dw.is_contiguous(memory_format=torch.channels_last)
    d = conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy)
    need = lib().gcc_conv_wgrad_workspace(C.byref(d))
    ws = workspace(need, x.device, 'wgrad')
    e0 = PROFILE.begin() if PROFILE.active else None
    check(lib().gcc_conv_wgrad(C.byref(d), xp, yp, dw.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel(),
                               stream()), 'gcc_conv_wgrad')
    if e0 is not None:
        PROFILE.end('wgrad_kernel (+ slab reduce)', 2.0 * N * Ho * Wo * Co * k * k * Ci, e0, shape=('wgrad', N, Ho, Wo, Ci, Co, k, stride))
    return dw


def wgrad_groupable(x, dy, dw, k, stride, pad):
    """would the library take this layer into a grouped weight gradient?  (gcc_conv_wgrad_group_workspace of the one-entry group:
    regular widths, no head / thin-output geometry)"""
    item = (_lib.wgrad_item_t * 1)()
    xp, N, Ci, H, W, ldx = geom(x)
    yp, _, Co, Ho, Wo, ldy = geom(dy)
    item[0].c = conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy)
    item[0].x, item[0].dy, item[0].dw, item[0].accumulate = xp, yp, dw.data_ptr(), 1
    return (dw.data_ptr() & 15) == 0 and lib().gcc_conv_wgrad_group_workspace(item, 1) > 0


class WgradGroup:
    """gcc_conv_wgrad_group_*: the weight gradients of several layers as ONE launch + one fold launch (include/gcc_hip.h).
    entries: [(x, dy, dw, k, stride, pad, accumulate)] with the operands of conv_wgrad.  The table is prepared on first use from the
    tensors' addresses and geometries and re-prepared whenever one of them changes (persistent activation / gradient buffers: never
    in steady state); the slab workspace and the device copy of the table belong to the object.  groupable(): False when the
    library refuses the set (an irregular width, a head / thin-output geometry) -- the caller then runs conv_wgrad per layer."""

    def __init__(self):
        self.key = None
        self.ok = False

    def _prepare(self, entries, key):
        n = len(entries)
        items = (_lib.wgrad_item_t * n)()
        flops = 0.0
        dev = entries[0][0].device
        for it, (x, dy, dw, k, stride, pad, acc) in zip(items, entries):
            xp, N, Ci, H, W, ldx = geom(x)
            yp, _, Co, Ho, Wo, ldy = geom(dy)
            assert tuple(dw.shape) == (Co, Ci, k, k) and dw.dtype == torch.float32
            assert k == 1 or dw.is_contiguous(memory_format=torch.channels_last)
            it.c = conv_desc(N, H, W, Ci, Co, k, stride, pad, ldx, ldy)
            it.x, it.dy, it.dw, it.accumulate = xp, yp, dw.data_ptr(), int(acc)
            flops += 2.0 * N * Ho * Wo * Co * k * k * Ci
        self.key, self.flops, self.n = key, flops, n
        need = lib().gcc_conv_wgrad_group_workspace(items, n) if n <= _lib.WGRAD_GROUP_MAX else 0
        self.ok = need > 0
        if not self.ok:
            return
        # a launch of the previous table may still be in flight on a stream the caching allocator does not track: the buffers it
        # reads stay alive for a few generations (re-preparing is rare: the operands are persistent buffers)
        if getattr(self, 'table_dev', None) is not None:
            self._retired = (getattr(self, '_retired', []) + [(self.ws, self.table_dev)])[-4:]
        self.ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        tb = int(lib().gcc_conv_wgrad_group_table_bytes())
        self.table_host = torch.zeros(tb, dtype=torch.uint8)
        check(lib().gcc_conv_wgrad_group_prepare(items, n, self.ws.data_ptr(), self.ws.numel(), self.table_host.data_ptr()),
              'gcc_conv_wgrad_group_prepare')
        self.table_dev = self.table_host.to(dev)          # (a synchronous copy, once per table)

    def groupable(self, entries):
        key = tuple((x.data_ptr(), dy.data_ptr(), dw.data_ptr(), tuple(x.shape), tuple(dy.shape), x.stride(), dy.stride(), k, s_, p_, bool(a))
                    for (x, dy, dw, k, s_, p_, a) in entries) + (tuple(sorted(current_plan().items())),)
        if key != self.key:
            self._prepare(entries, key)
        return self.ok

    def run(self):
        e0 = PROFILE.begin() if PROFILE.active else None
        check(lib().gcc_conv_wgrad_group_run(self.table_dev.data_ptr(), self.table_host.data_ptr(), stream()), 'gcc_conv_wgrad_group_run')
        if e0 is not None:
            PROFILE.end('wgrad_group_kernel (+ fold)', self.flops, e0, shape=('wgrad_group', self.n))


def seg_phys(n, split):
    """physical channel count of an n-channel dimension made of two 8-padded parts (split = first part)"""
    return ceil8(split) + ceil8(n - split) if 0 < split < n else ceil8(n)


def conv_wgrad_seg(x, dy, dw, rows, cols, row_split, col_split, k, stride, pad, accumulate=True):
    """weight gradient when x / dy carry concatenated (8-padded) channel parts; dw is the logical
    [rows, cols, k, k] channels_last gradient"""
    xp, N, Ci, H, W, ldx = geom(x)
    yp, _, Co, Ho, Wo, ldy = geom(dy)
    # the kernels work on the padded physical channel ranges (pad channels are zeros by the layout contract)
    Cip, Cop = seg_phys(cols, col_split), seg_phys(rows, row_split)
    assert Ci in (cols, Cip) and Co in (rows, Cop) and ldx >= Cip and ldy >= Cop, (Ci, Co, rows, cols, row_split, col_split)
    assert tuple(dw.shape) == (rows, cols, k, k) and dw.dtype == torch.float32
    d = conv_desc(N, H, W, Cip, Cop, k, stride, pad, ldx, ldy)
    ws = workspace(lib().gcc_conv_wgrad_workspace(C.byref(d)), x.device, 'wgrad')
    e0 = PROFILE.begin() if PROFILE.active else None
    check(lib().gcc_conv_wgrad_seg(C.byref(d), xp, yp, dw.data_ptr(), rows, cols, row_split, col_split, int(accumulate),
                                   ws.data_ptr(), ws.numel(), stream()), 'gcc_conv_wgrad_seg')
    if e0 is not None:
        PROFILE.end('wgrad_kernel (+ slab reduce)', 2.0 * N * Ho * Wo * rows * k * k * cols, e0,
                    shape=('wgrad', N, Ho, Wo, cols, rows, k, stride))
    return dw


def nchw_to_nhwc(src, dst, off=0, cfill=None):
    """src fp32 [N,C,H,W] contiguous -> dst NHWC bf16 channels [off, off+C) (+ zero fill to cfill)."""
    N, Cc, H, W = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous()
    dp, _, _, _, _, ld = geom(dst)
    check(lib().gcc_nchw_f32_to_nhwc_bf16(src.data_ptr(), dp, N, Cc, H, W, ld, off, cfill if cfill else Cc, stream()),
          'gcc_nchw_f32_to_nhwc_bf16')
    return dst


def nhwc_to_nchw(src, Cc=None):
    sp, N, C0, H, W, ld = geom(src)
    Cc = Cc or C0
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=src.device)
    check(lib().gcc_nhwc_bf16_to_nchw_f32(sp, out.data_ptr(), N, Cc, H, W, ld, 0, stream()), 'gcc_nhwc_bf16_to_nchw_f32')
    return out


def nhwc_copy(src, soff, dst, doff, Cc, cfill=None):
    sp, N, _, H, W, lds = geom(src)
    dp, _, _, _, _, ldd = geom(dst)
    check(lib().gcc_nhwc_copy(sp, lds, soff, dp, ldd, doff, Cc, cfill if cfill else Cc, N * H * W, stream()), 'gcc_nhwc_copy')


def nhwc_pack_pair(a, b, dst, Ca, Cb, doff=0):
    """dst[:, doff : doff + Ca + Cb] = cat(a[:, :Ca], b[:, :Cb]), zero-filled to the 8-wide group (Ca + Cb <= 8)"""
    ap, N, _, H, W, lda = geom(a)
    bp, _, _, _, _, ldb = geom(b)
    dp, _, _, _, _, ldd = geom(dst)
    check(lib().gcc_nhwc_pack_pair(ap, lda, 0, bp, ldb, 0, dp, ldd, doff, Ca, Cb, N * H * W, stream()), 'gcc_nhwc_pack_pair')


def nhwc_add(src, soff, dst, doff, Cc):
    sp, N, _, H, W, lds = geom(src)
    dp, _, _, _, _, ldd = geom(dst)
    check(lib().gcc_nhwc_add(sp, lds, soff, dp, ldd, doff, Cc, N * H * W, stream()), 'gcc_nhwc_add')


class BNState:
    """Per-application saved statistics of one BatchNorm layer (fp32 [C] each)."""

    def __init__(self, Cc, device):
        z = torch.zeros((4, Cc), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = z[0], z[1], z[2], z[3]


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, st, eps=1e-5, momentum=0.1):
    tiles, _, Cc = stats.shape
    check(lib().gcc_bn_finalize(stats.data_ptr(), tiles, Cc, float(count), gamma.data_ptr(), beta.data_ptr(), eps,
                                momentum, running_mean.data_ptr() if running_mean is not None else None,
                                running_var.data_ptr() if running_var is not None else None, st.mean.data_ptr(),
                                st.rstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(), stream()), 'gcc_bn_finalize')


def bn_eval_coeffs(gamma, beta, running_mean, running_var, st, eps=1e-5):
    check(lib().gcc_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                                   eps, gamma.numel(), st.scale.data_ptr(), st.shift.data_ptr(), stream()),
          'gcc_bn_eval_coeffs')


def _p(t):
    return t.data_ptr() if t is not None else None


def reflect_pad(src, dst, pad, backward=False):
    """forward: dst [N,C,H+2p,W+2p] = ReflectionPad2d(p)(src); backward: dst [N,C,H,W] = adjoint of it applied to src"""
    sp, N, Cc, Hs, Ws, lds = geom(src)
    dp, _, _, Hd, Wd, ldd = geom(dst)
    H, W = (Hd, Wd) if backward else (Hs, Ws)
    assert (Hs, Ws) == ((H + 2 * pad, W + 2 * pad) if backward else (H, W))
    check(lib().gcc_reflect_pad(sp, lds, dp, ldd, N, H, W, Cc, pad, int(backward), stream()), 'gcc_reflect_pad')


def dwconv_fwd(x, w, bias, out):
    xp, N, Cc, H, W, ldx = geom(x)
    op, _, _, _, _, ldo = geom(out)
    check(lib().gcc_dwconv3x3_reflect(0, xp, ldx, None, 0, op, ldo, w.data_ptr(), _p(bias), N, H, W, Cc, stream()),
          'gcc_dwconv3x3_reflect')


def dwconv_bwd_data(dy, w, out):
    yp, N, Cc, H, W, ldy = geom(dy)
    op, _, _, _, _, ldo = geom(out)
    check(lib().gcc_dwconv3x3_reflect(1, None, 0, yp, ldy, op, ldo, w.data_ptr(), None, N, H, W, Cc, stream()),
          'gcc_dwconv3x3_reflect')


def dwconv_wgrad(x, dy, dw, dbias):
    """dw [C,1,3,3] fp32 (+=), dbias [C] (+=)"""
    xp, N, Cc, H, W, ldx = geom(x)
    yp, _, _, _, _, ldy = geom(dy)
    ws = workspace(lib().gcc_dwconv3x3_wgrad_workspace(N, H, W, Cc), x.device, 'dwwgrad')
    check(lib().gcc_dwconv3x3_reflect_wgrad(xp, ldx, yp, ldy, dw.data_ptr(), _p(dbias), N, H, W, Cc, ws.data_ptr(), ws.numel(),
                                            stream()), 'gcc_dwconv3x3_reflect_wgrad')


class INState:
    """per-application InstanceNorm statistics: [N][C] mean / rstd / scale / shift"""

    def __init__(self, N, Cc, device):
        z = torch.zeros((4, N, Cc), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.shift = z[0], z[1], z[2], z[3]
        self.N = N
        self.ptrs = (z[0].data_ptr(), z[1].data_ptr(), z[2].data_ptr(), z[3].data_ptr())       # (the host's hot path: one tuple, no tensor calls)


def channel_stats(x):
    """per image partial sums of an activation in the conv-epilogue format [N][tiles][2][C]"""
    xp, N, Cc, H, W, ld = geom(x)
    tiles = lib().gcc_channel_stats_tiles(H * W, Cc)
    st = torch.empty((N, tiles, 2, Cc), dtype=torch.float32, device=x.device)
    check(lib().gcc_channel_stats(xp, ld, 0, Cc, H * W, N, st.data_ptr(), stream()), 'gcc_channel_stats')
    return st


def in_finalize(stats, count, st, eps=1e-5):
    """stats [N][tiles][2][C] -> per image mean / rstd / scale / shift"""
    N, tiles, _, Cc = stats.shape
    check(lib().gcc_in_finalize(stats.data_ptr(), tiles, N, Cc, float(count), eps, st.mean.data_ptr(), st.rstd.data_ptr(),
                                st.scale.data_ptr(), st.shift.data_ptr(), stream()), 'gcc_in_finalize')


INORM_FUSED_MAX_HW = int(os.environ.get('GCC_INORM_FUSED_MAX_HW', str(1 << 20)))   # planes above this take the three-pass route
INORM_WS_BYTES = 4096 + (3 << 19)      # include/gcc_hip.h: GCC_INORM_WORKSPACE_BYTES
_inorm_ws = {}
_zero_ws = {}


def zeroed_workspace(device, slot, nbytes):
    """zero-filled once, one per (slot, stream): for kernels that count arrivals in their workspace and leave the counter
    at zero (gcc_prelu's ordered slope-gradient sum)"""
    key = (device, slot, stream())
    ws = _zero_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        grow = max(int(nbytes), 2 * ws.numel() if ws is not None else 0)
        ws = _zero_ws[key] = torch.zeros(grow, dtype=torch.uint8, device=device)
    return ws


TAIL_WS_BYTES = 4096 + (4 << 20) + 4096 + (3 << 19)      # include/gcc_hip.h: GCC_TAIL_WORKSPACE_BYTES
_tail_ws = {}


def tail_workspace(device):
    """zero-filled once, one per stream (gcc_bn_t.tail_ws: the ticket words and group sums of the BatchNorm finalize that the
    last-arriving workgroups of a conv launch perform; the launches of one stream are ordered, so its layers share it)"""
    key = (device, stream())
    ws = _tail_ws.get(key)
    if ws is None:
        ws = _tail_ws[key] = torch.zeros(TAIL_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


FOLD_GRID = os.environ.get('GCC_FOLD_GRID', '1') != '0'      # the U-Net's split layers: fold + BatchNorm + activation as one full-chip kernel (GCC_OPT_FUSE_BN 3)
IN_CONV_FINALIZE = os.environ.get('GCC_IN_CONV_FINALIZE', '1') != '0'      # the conv launch's last-arriving workgroups finalize (0: a gcc_bn_finalize launch; profiles/r4_summary.md)


def bn_desc(bn_module, st, count, device, running=True):
    """gcc_bn_t of a training-mode BatchNorm2d application: statistics of `count` pixels, coefficients into the BNState `st`;
    running=False leaves running_mean / running_var alone (a pass that runs ahead of its place: engine.PatchGANEngine)"""
    # (bench.py's bracketed roofline step times every conv launch on its own: no finalize tail inside it there)
    ws = tail_workspace(device) if ((IN_CONV_FINALIZE or FOLD_GRID) and not PROFILE.active) else None
    return _lib.bn_t(bn_module.weight.data_ptr(), bn_module.bias.data_ptr(), bn_module.eps, bn_module.momentum, float(count),
                     bn_module.running_mean.data_ptr() if running else None, bn_module.running_var.data_ptr() if running else None,
                     st.mean.data_ptr(), st.rstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                     ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                     1 if (IN_CONV_FINALIZE and ws is not None) else 0, 0)


def inorm_workspace(device):
    """zero-filled once, one per stream (gcc_inorm_fwd's workspace contract: its in-launch barrier counts in it)"""
    key = (device, stream())
    ws = _inorm_ws.get(key)
    if ws is None:
        ws = _inorm_ws[key] = torch.zeros(INORM_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


_inorm_ws_raw = {}


def _inorm_ws_ptr(device, raw):
    """(pointer, bytes) of the stream's InstanceNorm workspace: the per-launch form of inorm_workspace"""
    got = _inorm_ws_raw.get(raw)
    if got is None:
        ws = inorm_workspace(device)
        got = _inorm_ws_raw[raw] = (ws.data_ptr(), ws.numel())
    return got


def _inorm_fwd_c(*a):
    global _inorm_fwd_c
    _inorm_fwd_c = lib().gcc_inorm_fwd           # bound once: the launch path then calls the ctypes function object directly
    return _inorm_fwd_c(*a)


def _inorm_bwd_c(*a):
    global _inorm_bwd_c
    _inorm_bwd_c = lib().gcc_inorm_bwd
    return _inorm_bwd_c(*a)


def inorm_fwd(x, y, st, act=ACT_NONE, slope=0.2, residual=None, eps=1e-5):
    """InstanceNorm2d(affine=False) + activation (+ residual) in one launch; st (INState) receives mean / rstd"""
    xp, N, Cc, H, W, ldx = geom(x)
    yp, _, _, _, _, ldy = geom(y)
    rp, ldr = (None, 0)
    if residual is not None:
        rp, _, _, _, _, ldr = geom(residual)
    raw = stream()
    wsp, wsb = _inorm_ws_ptr(x.device, raw)
    m, r, sc, sh = st.ptrs
    rc = _inorm_fwd_c(xp, ldx, yp, ldy, rp, ldr, Cc, H * W, N, act, slope, eps, m, r, sc, sh, wsp, wsb, raw)
    if rc:
        check(rc, 'gcc_inorm_fwd')


def inorm_bwd(x, y, g, dx, st, act=ACT_NONE, slope=0.2):
    xp, N, Cc, H, W, ldx = geom(x)
    yp, ldy = (None, 0)
    if y is not None:
        yp, _, _, _, _, ldy = geom(y)
    gp, _, _, _, _, ldg = geom(g)
    dxp, _, _, _, _, lddx = geom(dx)
    raw = stream()
    wsp, wsb = _inorm_ws_ptr(x.device, raw)
    rc = _inorm_bwd_c(xp, ldx, yp, ldy, gp, ldg, dxp, lddx, Cc, H * W, N, act, slope, st.ptrs[0], st.ptrs[1], wsp, wsb, raw)
    if rc:
        check(rc, 'gcc_inorm_bwd')


def bnact_fwd(x, y, y2=None, scale=None, shift=None, gate=None, gate_after_act=False, act=ACT_NONE, slope=0.2,
              act2=ACT_NONE, drop_p=0.0, seed=0, groups=1, residual=None):
    xp, N, Cc, H, W, ldx = geom(x)
    yp, ldy = (None, 0)
    if y is not None:
        yp, _, _, _, _, ldy = geom(y)
    y2p, ldy2 = (None, 0)
    if y2 is not None:
        y2p, _, _, _, _, ldy2 = geom(y2)
    rp, ldr = (None, 0)
    if residual is not None:
        rp, _, _, _, _, ldr = geom(residual)
    p = _lib.bnact_t(_p(scale), _p(shift), _p(gate), int(gate_after_act), act, slope, act2, drop_p, seed, groups, ldr, rp)
    pixels = N * H * W if groups <= 1 else H * W
    assert groups <= 1 or groups == N
    check(lib().gcc_bnact_fwd(C.byref(p), xp, ldx, 0, yp, ldy, 0, y2p, ldy2, 0, Cc, pixels, stream()), 'gcc_bnact_fwd')


BN_BWD_TAIL = os.environ.get('GCC_BN_BWD_TAIL', '0') == '1'      # 1: the reduce launch's last-arriving workgroups finalize (measured: profiles/r4_summary.md)
BN_BWD_GRID = os.environ.get('GCC_BN_BWD_GRID', '1') != '0'
# gcc_bn_bwd_one_launch_ex (the U-Net's layers): 0 off; 1 (default) tensors of <= BN_BWD_GRID_EX_MAX_PIXELS pixels; 2 every size
BN_BWD_GRID_EX = int(os.environ.get('GCC_BN_BWD_GRID_EX', '1'))
BN_BWD_GRID_EX_MAX_PIXELS = int(os.environ.get('GCC_BN_BWD_GRID_EX_MAX_PIXELS', '4096'))
BN_BWD_GRID_MIN_PIXELS = 4096        # at or below: bnact_bwd_small_kernel (one workgroup per 8 channels) is the one-launch form
# above: reduce + finalize + apply.  The one-launch kernel runs on the grid family's 4 x CUs / Q workgroups (one wave per SIMD): it wins
# while the tensor is a few launches' worth of latency and loses once it is bandwidth (profiles/r5_bn_bwd_paths.txt: 9.4 MB 23.7 against
# 25.7 us, 18.9 MB 44.4 against 36.6, 151 MB 388 against 204)
BN_BWD_GRID_MAX_BYTES = int(os.environ.get('GCC_BN_BWD_GRID_MAX_BYTES', str(12 << 20)))


def bnact_bwd(x, y, g1, dx, g2=None, bn=None, gamma=None, beta=None, bn_eval=False, gate=None, gate_after_act=False,
              act=ACT_NONE, slope=0.2, act2=ACT_NONE, drop_p=0.0, seed=0, dgamma=None, dbeta=None, dalpha=None,
              in_act=ACT_NONE, groups=1):
    """bn: BNState (training statistics) or None."""
    xp, N, Cc, H, W, ldx = geom(x)
    yp, ldy = (None, 0)
    if y is not None:
        yp, _, _, _, _, ldy = geom(y)
    g1p, _, _, _, _, ldg1 = geom(g1)
    g2p, ldg2 = (None, 0)
    if g2 is not None:
        g2p, _, _, _, _, ldg2 = geom(g2)
    dxp, _, _, _, _, lddx = geom(dx)
    pixels = N * H * W if groups <= 1 else H * W
    assert groups <= 1 or groups == N
    if (BN_BWD_GRID_EX and bn is not None and not bn_eval and gate is None and not gate_after_act and dalpha is None
            and in_act == ACT_NONE and groups <= 1 and not PROFILE.active
            and (g2 is not None or drop_p > 0.0 or (y is None and act != ACT_NONE) or pixels <= BN_BWD_GRID_MIN_PIXELS)
            and (BN_BWD_GRID_EX >= 2 or pixels <= BN_BWD_GRID_EX_MAX_PIXELS)):
        # the U-Net's BatchNorm backward (second gradient of the skip path, dropout, activation input recomputed from the forward's
        # affine) in ONE launch on the whole chip, small tensors included (round 4; gcc_bn_bwd_one_launch_ex)
        ws = inorm_workspace(x.device)
        rc = lib().gcc_bn_bwd_one_launch_ex(xp, ldx, yp, ldy, g1p, ldg1, g2p, ldg2, dxp, lddx, Cc, pixels, act, act2, slope, drop_p, seed,
                                            bn.mean.data_ptr(), bn.rstd.data_ptr(), bn.scale.data_ptr(), bn.shift.data_ptr(),
                                            _p(gamma), _p(dgamma), _p(dbeta), ws.data_ptr(), ws.numel(), stream())
        if rc == 0:
            return
        if rc != -2:
            check(rc, 'gcc_bn_bwd_one_launch_ex')
    if (BN_BWD_GRID and bn is not None and not bn_eval and g2 is None and gate is None and not gate_after_act and drop_p == 0.0
            and dalpha is None and in_act == ACT_NONE and act2 == ACT_NONE and groups <= 1 and (y is not None or act == ACT_NONE)
            and pixels > BN_BWD_GRID_MIN_PIXELS and pixels * ceil8(Cc) * 2 <= BN_BWD_GRID_MAX_BYTES and not PROFILE.active):
        # plain training-mode BatchNorm backward (SRResNet / SAGAN-generator blocks): one launch instead of reduce + finalize + apply
        ws = inorm_workspace(x.device)
        rc = lib().gcc_bn_bwd_one_launch(xp, ldx, yp, ldy, g1p, ldg1, dxp, lddx, Cc, pixels, act, slope, bn.mean.data_ptr(),
                                         bn.rstd.data_ptr(), _p(gamma), _p(dgamma), _p(dbeta), ws.data_ptr(), ws.numel(), stream())
        if rc == 0:
            return
        if rc != -2:                    # GCC_ERR_UNSUPPORTED: the geometry stays with the three-launch route below
            check(rc, 'gcc_bn_bwd_one_launch')
    p = _lib.bnact_bwd_t(1 if bn is not None else 0, int(bn_eval), _p(bn.mean) if bn is not None else None,
                         _p(bn.rstd) if bn is not None else None, _p(gamma), _p(beta), _p(gate), int(gate_after_act),
                         act, slope, act2, drop_p, seed, _p(dgamma), _p(dbeta), _p(dalpha), groups, 1 if BN_BWD_TAIL else 0)
    need = lib().gcc_bnact_bwd_workspace(Cc, pixels) * max(1, groups)
    # zero-filled once, one per stream (flags bit 0: the reduce launch's last-arriving workgroups do the finalize step)
    ws = zeroed_workspace(x.device, 'bnbwd', need) if BN_BWD_TAIL else workspace(need, x.device, 'bnbwd')
    e0 = PROFILE.begin() if (PROFILE.active and PROFILE.streaming) else None
    check(lib().gcc_bnact_bwd_ex(C.byref(p), in_act, slope, xp, ldx, 0, yp, ldy, 0, g1p, ldg1, 0, g2p, ldg2, 0, dxp, lddx,
                                 0, Cc, pixels, ws.data_ptr(), ws.numel(), stream()), 'gcc_bnact_bwd')
    if e0 is not None:          # "flops" carries bytes here: 8 B/elem reduce (+6 apply) with training BN, 6 without
        PROFILE.end('bnact_bwd (bytes)', float(N * H * W * ceil8(Cc)) * (14.0 if (bn is not None and not bn_eval) else 6.0), e0,
                    shape=('bnbwd', N, Cc, H, W, 'bn' if bn is not None else 'act', 'gate' if gate is not None else ''))


def channel_sum(x, out, accumulate=False):
    xp, N, Cc, H, W, ld = geom(x)
    need = lib().gcc_channel_sum_workspace(Cc, N * H * W)
    ws = workspace(need, x.device, 'chansum')
    check(lib().gcc_channel_sum(xp, ld, 0, Cc, N * H * W, out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel(),
                                stream()), 'gcc_channel_sum')


def channel_sum_group(pairs, accumulate=True):
    """[(x, out)]: out (+)= per-channel sums of x, every tensor of <= 16384 pixels in ONE launch per 24 (gcc_channel_sum_group);
    larger ones through channel_sum"""
    small = []
    for x, out in pairs:
        xp, N, Cc, H, W, ld = geom(x)
        if N * H * W <= _lib.CHANSUM_SMALL_MAX_PIXELS:
            small.append((xp, ld, Cc, N * H * W, out.data_ptr()))
        else:
            channel_sum(x, out, accumulate=accumulate)
    for i in range(0, len(small), _lib.CHANSUM_GROUP_MAX):
        part = small[i:i + _lib.CHANSUM_GROUP_MAX]
        items = (_lib.chansum_item_t * len(part))()
        for it, (xp, ld, Cc, px, op) in zip(items, part):
            it.x, it.ld, it.off, it.C, it.pixels, it.out, it.accumulate = xp, ld, 0, Cc, px, op, int(accumulate)
        check(lib().gcc_channel_sum_group(items, len(part), stream()), 'gcc_channel_sum_group')


def gate_mask(alpha, tau, mask):
    check(lib().gcc_gate_mask(alpha.data_ptr(), float(tau), mask.data_ptr(), alpha.numel(), stream()), 'gcc_gate_mask')


GAN_MODES = {'hinge': 0, 'lsgan': 1, 'vanilla': 2, 'wgangp': 3}


def gan_loss(mode, pred, target_is_real, for_discriminator, loss, dpred=None, grad_weight=1.0, weight_dev=None,
             dpred_accumulate=False):
    """loss[0] = GANLoss value ; dpred (+)= grad_weight * (*weight_dev) * dL/dpred"""
    pp, N, Cc, H, W, ld = geom(pred)
    assert Cc == 1
    dp = None
    if dpred is not None:
        dp, _, _, _, _, ldd = geom(dpred)
        assert ldd == ld
    check(lib().gcc_gan_loss_ex(GAN_MODES[mode], int(target_is_real), int(for_discriminator), pp, ld, 0, N * H * W,
                                loss.data_ptr(), dp, float(grad_weight), _p(weight_dev), int(dpred_accumulate), stream()),
          'gcc_gan_loss')


def spectral_power_iteration(w_bar, u, v, t_out, sigma_out, w_eff, slot='sn_fwd'):
    """one power iteration on the fp32 master (updates u, v in place), sigma and W_eff = W_bar / sigma"""
    R, Cc, kh, kw = w_bar.shape
    ws = workspace(lib().gcc_spectral_workspace(R, Cc, kh * kw), w_bar.device, slot)
    check(lib().gcc_spectral_power_iteration(w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), R, Cc, kh * kw, t_out.data_ptr(),
                                             sigma_out.data_ptr(), w_eff.data_ptr(), ws.data_ptr(), ws.numel(), stream()),
          'gcc_spectral_power_iteration')


SN_FUSED_PACK = os.environ.get('GCC_SN_FUSED_PACK', '1') != '0'


def spectral_power_iteration_pack(w_bar, u, v, t_out, sigma_out, w, wt, slot='sn_fwd'):
    """the power iteration with W_bar / sigma written straight into the bf16 packings w / wt (4 launches instead of 7)"""
    R, Cc, kh, kw = w_bar.shape
    ws = workspace(lib().gcc_spectral_workspace(R, Cc, kh * kw), w_bar.device, slot)
    check(lib().gcc_spectral_power_iteration_pack(w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), R, Cc, kh * kw, t_out.data_ptr(),
                                                  sigma_out.data_ptr(), w.data_ptr(), wt.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream()), 'gcc_spectral_power_iteration_pack')


def spectral_power_iteration_pack_group(entries, slot='sn_group'):
    """[(w_bar, u, v, t_out, sigma_out, w, wt)]: spectral_power_iteration_pack of every entry, all of them in four launches per
    SPECTRAL_GROUP_MAX layers (gcc_spectral_power_iteration_pack_group: the same bits)"""
    for i in range(0, len(entries), _lib.SPECTRAL_GROUP_MAX):
        part = entries[i:i + _lib.SPECTRAL_GROUP_MAX]
        items = (_lib.sn_item_t * len(part))()
        for it, (w_bar, u, v, t_out, sigma_out, w, wt) in zip(items, part):
            R, Cc, kh, kw = w_bar.shape
            it.w_bar, it.u, it.v, it.R, it.C, it.T = w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), R, Cc, kh * kw
            it.t_out, it.sigma_out, it.w, it.wt = t_out.data_ptr(), sigma_out.data_ptr(), w.data_ptr(), wt.data_ptr()
        ws = workspace(lib().gcc_spectral_group_workspace(items, len(part)), part[0][0].device, slot)
        check(lib().gcc_spectral_power_iteration_pack_group(items, len(part), ws.data_ptr(), ws.numel(), stream()),
              'gcc_spectral_power_iteration_pack_group')


def spectral_grad(g_eff, w_bar, u, v, t_fwd, sigma_fwd, dw_bar, du=None, dv=None, slot='sn_bwd'):
    R, Cc, kh, kw = w_bar.shape
    ws = workspace(lib().gcc_spectral_workspace(R, Cc, kh * kw), w_bar.device, slot)
    check(lib().gcc_spectral_grad(g_eff.data_ptr(), w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), t_fwd.data_ptr(),
                                  sigma_fwd.data_ptr(), R, Cc, kh * kw, dw_bar.data_ptr(), _p(du), _p(dv), ws.data_ptr(),
                                  ws.numel(), stream()), 'gcc_spectral_grad')


def attention_fwd(qkv, offs, x, gamma, Cc, C8, y, o, stats, A=None):
    """qkv: NHWC bf16 buffer holding q | k | v at channel offsets offs; y = gamma * softmax(q^T k) v + x.
    stats: fp32 [B, N, 2] saved for backward; A: optional fp32 [B, N, N] attention map"""
    qp, B, _, H, W, ldq = geom(qkv)
    xp, _, _, _, _, ldx = geom(x)
    yp, _, _, _, _, ldy = geom(y)
    op, _, _, _, _, ldo = geom(o)
    check(lib().gcc_attention_fwd(qp, ldq, offs[0], offs[1], offs[2], xp, ldx, gamma.data_ptr(), B, H * W, Cc, C8, yp, ldy,
                                  op, ldo, stats.data_ptr(), _p(A), stream()), 'gcc_attention_fwd')


def attention_bwd(qkv, offs, o, stats, gamma, dy, Cc, C8, dqkv, rowdot, dgamma=None):
    qp, B, _, H, W, ldq = geom(qkv)
    op, _, _, _, _, ldo = geom(o)
    dyp, _, _, _, _, lddy = geom(dy)
    dqp, _, _, _, _, lddq = geom(dqkv)
    assert lddq == ldq
    check(lib().gcc_attention_bwd(qp, ldq, offs[0], offs[1], offs[2], op, ldo, stats.data_ptr(), gamma.data_ptr(), dyp, lddy, B,
                                  H * W, Cc, C8, dqp, lddq, rowdot.data_ptr(), _p(dgamma), stream()), 'gcc_attention_bwd')


def arch_coeffs(Lfr, Lf, Lr, dT, loss, c_fr, c_f, weight=0.5):
    check(lib().gcc_arch_coeffs(Lfr.data_ptr(), Lf.data_ptr(), Lr.data_ptr(), dT.data_ptr(), float(weight), loss.data_ptr(),
                                c_fr.data_ptr(), c_f.data_ptr(), stream()), 'gcc_arch_coeffs')


def l1_loss(a, b, loss, weight=1.0, accumulate=False, da=None):
    ap, N, Cc, H, W, lda = geom(a)
    bp, _, _, _, _, ldb = geom(b)
    dap, ldda = (None, 0)
    if da is not None:
        dap, _, _, _, _, ldda = geom(da)
    ws = workspace(lib().gcc_loss_workspace(N * H * W, Cc), a.device, 'loss')
    check(lib().gcc_l1_loss(ap, lda, 0, bp, ldb, 0, Cc, N * H * W, float(weight), loss.data_ptr(), int(accumulate), dap,
                            ldda, 0, ws.data_ptr(), ws.numel(), stream()), 'gcc_l1_loss')


def mse_loss(a, b, loss, weight=1.0, accumulate=False, da=None):
    """loss (+)= weight * mean((a-b)^2); da = its gradient"""
    ap, N, Cc, H, W, lda = geom(a)
    bp, _, _, _, _, ldb = geom(b)
    dap, ldda = (None, 0)
    if da is not None:
        dap, _, _, _, _, ldda = geom(da)
    ws = workspace(lib().gcc_loss_workspace(N * H * W, Cc), a.device, 'loss')
    check(lib().gcc_mse_loss(ap, lda, 0, bp, ldb, 0, Cc, N * H * W, float(weight), loss.data_ptr(), int(accumulate), dap,
                             ldda, 0, ws.data_ptr(), ws.numel(), stream()), 'gcc_mse_loss')


def prelu_fwd(x, slope, y, shuffle=1):
    """y = prelu(x) (shuffle 1) or prelu(pixel_shuffle(x, 2)): x [N,4C,H,W] -> y [N,C,2H,2W]"""
    xp, N, Cx, H, W, ldx = geom(x)
    yp, _, Cy, _, _, ldy = geom(y)
    check(lib().gcc_prelu(0, xp, ldx, slope.data_ptr(), Cy, N, H, W, shuffle, yp, ldy, None, 0, None, 0, None, None, 0, stream()),
          'gcc_prelu')


def prelu_bwd(x, slope, dy, dx, dslope=None, shuffle=1):
    xp, N, Cx, H, W, ldx = geom(x)
    dyp, _, Cy, _, _, lddy = geom(dy)
    dxp, _, _, _, _, lddx = geom(dx)
    ws = zeroed_workspace(x.device, 'prelu', 256 + 4 * 4096) if dslope is not None else None
    check(lib().gcc_prelu(1, xp, ldx, slope.data_ptr(), Cy, N, H, W, shuffle, None, 0, dyp, lddy, dxp, lddx, _p(dslope),
                          _p(ws), ws.numel() if ws is not None else 0, stream()), 'gcc_prelu')


def maxpool_fwd(x, y):
    xp, N, Cc, H, W, ldx = geom(x)
    yp, _, _, Ho, Wo, ldy = geom(y)
    assert (H, W) == (2 * Ho, 2 * Wo)
    check(lib().gcc_maxpool2x2(0, xp, ldx, yp, ldy, None, 0, None, 0, N, Ho, Wo, Cc, stream()), 'gcc_maxpool2x2')


def maxpool_bwd(x, dy, dx, relu_mask=False):
    """relu_mask: x is a ReLU's output and dx is to be the gradient w.r.t. the ReLU's INPUT (its mask applied on the way)"""
    xp, N, Cc, H, W, ldx = geom(x)
    dyp, _, _, Ho, Wo, lddy = geom(dy)
    dxp, _, _, _, _, lddx = geom(dx)
    check(lib().gcc_maxpool2x2(2 if relu_mask else 1, xp, ldx, None, 0, dyp, lddy, dxp, lddx, N, Ho, Wo, Cc, stream()), 'gcc_maxpool2x2')


def pool_linear_fwd(x, w, b, pooled, logit):
    xp, N, Cc, H, W, ldx = geom(x)
    lp, _, _, _, _, ldl = geom(logit)
    check(lib().gcc_pool_linear_fwd(xp, ldx, N, H * W, Cc, w.data_ptr(), b.data_ptr(), pooled.data_ptr(), lp, ldl, stream()),
          'gcc_pool_linear_fwd')


def pool_linear_bwd(dlogit, w, pooled, like, dx=None, dw=None, db=None):
    """like: the pooled activation tensor (geometry); dx (same geometry) / dw (+=) / db (+=) are optional"""
    _, N, Cc, H, W, _ = geom(like)
    dlp, _, _, _, _, ldl = geom(dlogit)
    dxp, lddx = (None, 0)
    if dx is not None:
        dxp, _, _, _, _, lddx = geom(dx)
    check(lib().gcc_pool_linear_bwd(dlp, ldl, w.data_ptr(), pooled.data_ptr(), N, H * W, Cc, dxp, lddx, _p(dw), _p(db), stream()),
          'gcc_pool_linear_bwd')


def distill_workspace_bytes(N, Cc, HW):
    return lib().gcc_distill_workspace(N, Cc, HW)


def distill_fwd(f, t, out2, ws, squared=False):
    """out2 = (gram term, content term): RMSE form (Pix2Pix) or, squared, plain MSE (CycleGAN)"""
    fp, N, Cc, H, W, ldf = geom(f)
    tp, _, _, _, _, ldt = geom(t)
    check(lib().gcc_distill_fwd(fp, ldf, 0, tp, ldt, 0, N, Cc, H * W, int(squared), out2.data_ptr(), ws.data_ptr(),
                                ws.numel(), stream()), 'gcc_distill_fwd')


def distill_bwd(f, t, wg, wc, df, ws, squared=False):
    fp, N, Cc, H, W, ldf = geom(f)
    tp, _, _, _, _, ldt = geom(t)
    dp, _, _, _, _, ldd = geom(df)
    check(lib().gcc_distill_bwd(fp, ldf, 0, tp, ldt, 0, N, Cc, H * W, int(squared), float(wg), float(wc), dp, ldd, 0,
                                ws.data_ptr(), ws.numel(), stream()), 'gcc_distill_bwd')


class AdamPlan:
    """Device-side descriptor/work lists of one optimizer (built once; pointers must stay valid)."""
    CHUNK = 1 << 12      # elements per 256-thread workgroup: 16 per thread, enough workgroups to fill the chip for 10 M-parameter groups

    def __init__(self, params, grads, device, l1=None):
        self.params = list(params)
        self.grads = list(grads)
        self.m = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        self.v = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        self.step_count = 0
        self.last_hyper = None              # (lr, beta1, beta2, eps) of the last step
        self.device = device
        self.l1 = list(l1) if l1 is not None else [0.0] * len(self.params)
        self.grad_scale = 1.0
        self._build()

    def _build(self):
        n = len(self.params)
        T = (_lib.adam_tensor_t * n)()
        chunks = []
        for i, (p, g, m, v) in enumerate(zip(self.params, self.grads, self.m, self.v)):
            assert p.dtype == torch.float32 and g.dtype == torch.float32
            assert p.stride() == g.stride() == m.stride() == v.stride(), 'p/g/m/v must share a layout'
            T[i] = _lib.adam_tensor_t(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                                      float(self.l1[i]), float(self.grad_scale))
            for o in range(0, p.numel(), self.CHUNK):
                chunks.append((i, o))
        K = (_lib.adam_chunk_t * len(chunks))()
        for j, (i, o) in enumerate(chunks):
            K[j] = _lib.adam_chunk_t(i, 0, o)
        self.nchunks = len(chunks)
        self.d_tensors = torch.frombuffer(bytearray(bytes(T)), dtype=torch.uint8).to(self.device)
        self.d_chunks = torch.frombuffer(bytearray(bytes(K)), dtype=torch.uint8).to(self.device)

    def set_grad_scale(self, s):
        if s != self.grad_scale:
            self.grad_scale = s
            self._build()

    def step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.step_count += 1
        self.last_hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps))
        note_dynamic(self)
        check(lib().gcc_adam_step(self.d_tensors.data_ptr(), self.d_chunks.data_ptr(), self.nchunks, self.CHUNK,
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), self.step_count, stream()),
              'gcc_adam_step')


def _adam_replay_update(self, rec, tag):
    """one more step of this plan inside a replayed iteration: the launch's bias_correction1 / sqrt(bias_correction2)
    arguments (7 and 8 of adam_kernel, include/gcc_hip.h) for the new step count; lr / betas / eps are the recorded ones"""
    self.step_count += 1
    _, b1, b2, _ = self.last_hyper
    f = (C.c_float * 2)()
    check(lib().gcc_adam_factors(b1, b2, self.step_count, f), 'gcc_adam_factors')
    for idx in (0, 1):
        v = C.c_float(f[idx])
        n = lib().gcc_replay_patch(rec, tag, 7 + idx, C.byref(v), 4)
        if n != 1:
            raise RuntimeError('gcc_replay_patch(adam tag %d): %d launches patched' % (tag, n))


AdamPlan.replay_update = _adam_replay_update


def write_i32(dst, values):
    """dst (device int32) [0:len(values)] = values, carried by the launch itself (<= 16)"""
    arr = (C.c_int * len(values))(*values)
    check(lib().gcc_write_i32(dst.data_ptr(), arr, len(values), stream()), 'gcc_write_i32')


def image_pool_query(images, out, pool, sel):
    N, _, H, W = images.shape
    ip, _, _, _, _, ldi = geom(images)
    op_, _, _, _, _, ldo = geom(out)
    assert ldi == 8 and ldo == 8 and pool.stride(1) == 1 and pool.shape[1] <= 8
    check(lib().gcc_image_pool_query(ip, op_, pool.data_ptr(), sel.data_ptr(), N, H * W, pool.shape[0], stream()),
          'gcc_image_pool_query')


def fill(t, v):
    check(lib().gcc_fill_f32(t.data_ptr(), float(v), t.numel(), stream()), 'gcc_fill_f32')


def add_f32_(dst, src):
    assert dst.dtype == src.dtype == torch.float32 and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
    check(lib().gcc_add_f32(dst.data_ptr(), src.data_ptr(), dst.numel(), stream()), 'gcc_add_f32')


def clamp_(t, lo, hi):
    check(lib().gcc_clamp_f32(t.data_ptr(), float(lo), float(hi), t.numel(), stream()), 'gcc_clamp_f32')


def scalar_op(op, a, b, out, c=None, k0=0.0, k1=0.0):
    check(lib().gcc_scalar_op(op, a.data_ptr(), b.data_ptr(), _p(c), float(k0), float(k1), out.data_ptr(), stream()),
          'gcc_scalar_op')
