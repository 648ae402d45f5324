"""CPU-side checks: the C-ABI library loads and exports every symbol include/gcc_hip.h declares
(no compute calls without a GPU), the flag surface matches the reference's parsed options (golden),
the parameter-owning module trees have the reference's state_dict keys, and the product refuses to
run without its HIP library / GPU instead of falling back."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'gcc_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = set(re.findall(r'\b(gcc_[a-z0-9_]+)\s*\(', src))
    names.discard('gcc_conv_out')          # static inline helper
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from gcc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), 'libgcc_hip.so does not export %s' % name
        assert name in _lib.PROTOTYPES, 'no ctypes prototype for %s' % name
    for name in _lib.PROTOTYPES:
        assert name in decl, '%s bound but not declared in include/gcc_hip.h' % name
    _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'gcc_hip.h')).read()
    abi = int(re.search(r'#define\s+GCC_HIP_ABI\s+(\d+)', hdr).group(1))
    assert _lib.load().gcc_version() == abi == _lib.GCC_HIP_ABI
    assert b'workspace' in _lib.load().gcc_strerror(-3)


def test_loader_refuses_a_library_of_another_abi_generation(monkeypatch):
    """a stale libgcc_hip.so (or one named by GCC_HIP_LIB) whose struct layouts / option ids differ must not load"""
    from gcc_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'GCC_HIP_ABI', _lib.GCC_HIP_ABI + 1)
    with pytest.raises(_lib.GccError, match='GCC_HIP_ABI'):
        _lib.load()
    monkeypatch.undo()
    _lib.load()


def test_thin_output_route_rejects_more_than_eight_output_channels():
    """Co * KW <= 32 admits 3 x 3 layers with 9 or 10 output channels (SRGAN's pruned residual blocks, 24 -> 9 /
    64 -> 10), but the thin-output weight- and data-gradient kernels stage ONE 8-channel chunk of dY per pixel: such layers must take
    the generic kernels (route 0 / the ring walk), and the weight gradient must ask for no thin-output workspace"""
    from gcc_amd import _lib
    lib = _lib.load()
    conv = lambda N, H, W, Ci, Co, k, s, p: _lib.conv_t(N, H, W, Ci, Co, k, k, s, p, (Ci + 7) // 8 * 8, 0, (Co + 7) // 8 * 8, 0)
    none = _lib.epilogue_t(None, 0, 0.2, None, None, 0)
    route = lambda d, dgrad=0: lib.gcc_conv_route(ctypes.byref(d), dgrad, ctypes.byref(none))
    assert route(conv(2, 48, 48, 64, 3, 9, 1, 4)) == 3 and route(conv(2, 48, 48, 64, 3, 9, 1, 4), 1) == 3
    assert route(conv(2, 48, 48, 24, 8, 3, 1, 1), 1) == 3                        # 8 channels of dY: one chunk, still thin-output
    for ci, co in ((24, 9), (64, 10), (24, 10), (64, 9)):
        d = conv(2, 48, 48, ci, co, 3, 1, 1)
        assert route(d, 0) != 3 and route(d, 1) != 3, (ci, co)


def test_comm_entry_points_without_a_gpu():
    """gcc_comm_*: the id comes from RCCL (resolved at run time, no load-time dependency), argument errors are codes"""
    from gcc_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(128)
    assert lib.gcc_comm_unique_id(buf) == 0
    assert any(bytes(buf))
    assert lib.gcc_comm_unique_id(None) == -1
    h = ctypes.c_void_p()
    assert lib.gcc_comm_init(ctypes.byref(h), 2, 2, bytes(buf)) == -1          # rank out of range
    assert lib.gcc_comm_init(ctypes.byref(h), 0, 0, bytes(buf)) == -1
    assert lib.gcc_comm_init(None, 0, 1, bytes(buf)) == -1
    assert lib.gcc_comm_allreduce_sum_f32(None, None, 0, None) == -1
    assert lib.gcc_comm_destroy(None) == -1
    assert lib.gcc_comm_rank(None) == -1 and lib.gcc_comm_world(None) == -1 and lib.gcc_comm_count(None) == -1
    import subprocess
    out = subprocess.run(['readelf', '-d', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'rccl' not in out, 'libgcc_hip.so must not depend on RCCL at load time'


def test_replay_recorder_lifecycle_without_a_gpu():
    """gcc_replay_* host state (no launch is made): one recording per thread, an empty recording closes, reports and replays
    as nothing; patching needs a closed recording and a positive tag; gcc_adam_factors is the arithmetic gcc_adam_step uses"""
    import math
    from gcc_amd import _lib
    lib = _lib.load()
    h, h2 = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.gcc_replay_begin(ctypes.byref(h)) == 0 and h.value
    assert lib.gcc_replay_begin(ctypes.byref(h2)) != 0           # this thread is already recording
    assert lib.gcc_replay_tag_next(5) == 0
    v = ctypes.c_float(1.0)
    assert lib.gcc_replay_patch(h, 5, 0, ctypes.byref(v), 4) < 0  # still open
    assert lib.gcc_replay_end(h, 4) == 0
    assert lib.gcc_replay_end(h, 4) != 0                          # closed already
    assert [lib.gcc_replay_info(h, i) for i in range(5)] == [0, 0, 0, 1, 0]
    assert lib.gcc_replay_patch(h, 5, 0, ctypes.byref(v), 4) == 0 # nothing carries the tag
    assert lib.gcc_replay_patch(h, 0, 0, ctypes.byref(v), 4) < 0
    assert lib.gcc_replay_run(h) == 0
    assert lib.gcc_replay_destroy(h) == 0
    assert lib.gcc_replay_begin(ctypes.byref(h)) == 0             # a new recording may start
    assert lib.gcc_replay_destroy(h) == 0                         # ... and be dropped unfinished
    assert lib.gcc_replay_begin(ctypes.byref(h)) == 0 and lib.gcc_replay_end(h, 1) == 0 and lib.gcc_replay_destroy(h) == 0
    f = (ctypes.c_float * 2)()
    for step in (1, 2, 10, 1000):
        assert lib.gcc_adam_factors(0.5, 0.999, step, f) == 0
        b2 = float(np.float32(0.999))
        assert f[0] == np.float32(1.0 - 0.5 ** step) and f[1] == np.float32(math.sqrt(1.0 - b2 ** step))
    assert lib.gcc_adam_factors(0.5, 0.999, 0, f) != 0


def test_weight_gradient_plan_follows_the_schedule(monkeypatch):
    """models/_streams.py + ops.set_plan: every model instance keeps the tile plan its schedule asked for and states it at the
    head of its phases, for the thread that enqueues: the production plan halves the workgroup targets of split weight-gradient
    launches and leaves the pair split off, the alone plan keeps the library's defaults with the pair split (and the 128-column
    halo tiles) on; an explicit GCC_WGRAD_WGS* environment value wins; two instances with different plans each get their own
    when their turn comes; another THREAD never sees either (the plan travels in gcc_conv_t.plan, round 5: no library state)"""
    import threading
    from gcc_amd import ops
    from gcc_amd.models._streams import TeacherStreamMixin as M
    monkeypatch.delenv('GCC_PAIR_CONCURRENT', raising=False)
    monkeypatch.setattr(ops, '_plan_pinned', {})

    class Fake(M):
        device = None
        serialize_streams = True

    def state():
        d = ops.conv_desc(16, 32, 32, 512, 1024, 4, 1, 1, 512, 1024)
        return (d.plan.pair, d.plan.halo_hc, d.plan.wgrad_wgs_big, d.plan.wgrad_wgs)
    a, b = Fake(), Fake()
    try:
        a.set_stream_schedule(True)
        assert state() == (0, 0, 128, 256)
        b.set_stream_schedule(False)
        assert state() == (1, 1, 0, 0)
        a._ensure_plan()                         # head of a's next phase (_teacher_stream): its own plan again, whatever b left
        assert state() == (0, 0, 128, 256)
        seen = []
        t = threading.Thread(target=lambda: seen.append(state()))
        t.start()
        t.join()
        assert seen == [(0, 0, 0, 0)]            # a thread that stated nothing launches with the library's defaults
        b._ensure_plan()
        assert state() == (1, 1, 0, 0)
        with ops.plan_override(halo_hc=128):
            assert state() == (1, 128, 0, 0)
        assert state() == (1, 1, 0, 0)
        a.set_stream_schedule(False, plan='production')      # one stream under the production plan (bench.py's bracketed step)
        assert state() == (0, 0, 128, 256)
        monkeypatch.setattr(ops, '_plan_pinned', {'wgrad_wgs': 512})       # GCC_WGRAD_WGS=512 in the environment
        a.set_stream_schedule(True)
        assert state() == (0, 0, 128, 512)
    finally:
        a.restore_library_plan()
    monkeypatch.setattr(ops, '_plan_pinned', {})
    ops.set_plan()
    assert state() == (0, 0, 0, 0)


def test_grouped_weight_gradient_host_logic():
    """gcc_conv_wgrad_group_* (round 6) without a GPU: which entries the library takes into a group, the workspace / table sizes,
    argument errors.  (prepare only writes host memory; the device pointers are addresses here)"""
    from gcc_amd import _lib
    lib = _lib.load()
    conv = lambda N, H, W, Ci, Co, k, s, p: _lib.conv_t(N, H, W, Ci, Co, k, k, s, p, (Ci + 7) // 8 * 8, 0, (Co + 7) // 8 * 8, 0)

    def items(descs, dw=0x40000000):
        arr = (_lib.wgrad_item_t * len(descs))()
        for it, d in zip(arr, descs):
            it.c, it.x, it.dy, it.dw, it.accumulate = d, 0x10000000, 0x20000000, dw, 1
        return arr
    unet = [conv(16, 128, 128, 32, 64, 4, 2, 1), conv(16, 64, 64, 64, 128, 4, 2, 1), conv(16, 2, 2, 256, 256, 4, 2, 1),
            conv(16, 16, 16, 128, 512, 4, 2, 1)]
    tb = lib.gcc_conv_wgrad_group_table_bytes()
    assert tb > 1024
    ws = lib.gcc_conv_wgrad_group_workspace(items(unet), len(unet))
    assert ws > 0
    table = ctypes.create_string_buffer(tb)
    assert lib.gcc_conv_wgrad_group_prepare(items(unet), len(unet), ctypes.c_void_p(0x50000000), ws, table) == 0
    assert any(bytes(table))
    assert lib.gcc_conv_wgrad_group_prepare(items(unet), len(unet), ctypes.c_void_p(0x50000000), ws - 1, table) == -3      # workspace
    assert lib.gcc_conv_wgrad_group_prepare(items(unet), len(unet), None, ws, table) == -1
    assert lib.gcc_conv_wgrad_group_prepare(items(unet, dw=0x40000004), len(unet), ctypes.c_void_p(0x50000000), ws, table) == -1   # dw not 16-byte aligned
    # entries that keep their own route: a 3-channel image layer, the single-output-channel head, a thin-output 9 x 9 layer
    for bad in (conv(16, 256, 256, 3, 32, 4, 2, 1), conv(16, 31, 31, 1024, 1, 4, 1, 1), conv(2, 96, 96, 64, 3, 9, 1, 4)):
        assert lib.gcc_conv_wgrad_group_workspace(items(unet + [bad]), len(unet) + 1) == 0
        assert lib.gcc_conv_wgrad_group_prepare(items(unet + [bad]), len(unet) + 1, ctypes.c_void_p(0x50000000), 1 << 30, table) == -2
    assert lib.gcc_conv_wgrad_group_workspace(items(unet), 0) == 0
    assert lib.gcc_conv_wgrad_group_workspace(items(unet * 9), 36) == 0                      # more than GCC_WGRAD_GROUP_MAX entries
    assert lib.gcc_conv_wgrad_group_run(None, table, None) == -1
    empty = ctypes.create_string_buffer(tb)
    assert lib.gcc_conv_wgrad_group_run(ctypes.c_void_p(0x60000000), empty, None) == -1     # not a prepared table
    # gcc_channel_sum_group: argument errors are codes
    cs = (_lib.chansum_item_t * 1)()
    assert lib.gcc_channel_sum_group(cs, 0, None) == -1 and lib.gcc_channel_sum_group(None, 1, None) == -1
    cs[0].x, cs[0].ld, cs[0].off, cs[0].C, cs[0].pixels, cs[0].out, cs[0].accumulate = 0x10000000, 64, 0, 64, 1 << 20, 0x20000000, 1
    assert lib.gcc_channel_sum_group(cs, 1, None) == -2                                       # more than 16384 pixels: gcc_channel_sum's job


def test_conv_route_predicates():
    """gcc_conv_route is host logic only: which kernel family each layer shape of the headline config runs on"""
    from gcc_amd import _lib
    lib = _lib.load()
    conv = lambda N, H, W, Ci, Co, k, s, p: _lib.conv_t(N, H, W, Ci, Co, k, k, s, p, (Ci + 7) // 8 * 8, 0, (Co + 7) // 8 * 8, 0)
    none = _lib.epilogue_t(None, 0, 0.2, None, None, 0)
    d_l1 = conv(16, 256, 256, 6, 128, 4, 2, 1)
    assert lib.gcc_conv_route(ctypes.byref(d_l1), 0, ctypes.byref(none)) == 1          # thin fprop
    assert lib.gcc_conv_route(ctypes.byref(d_l1), 1, ctypes.byref(none)) == 1          # 128 channels in: the LDS-staged thin kernel
    prev = lib.gcc_set_option(_lib.OPT_IGEMM_THIN, 2)
    try:
        assert lib.gcc_conv_route(ctypes.byref(d_l1), 1, ctypes.byref(none)) == 0      # ... or the implicit GEMM without it
        assert lib.gcc_conv_route(ctypes.byref(d_l1), 0, ctypes.byref(none)) == 1
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, prev)
    g_u0 = conv(16, 256, 256, 3, 64, 4, 2, 1)
    assert lib.gcc_conv_route(ctypes.byref(g_u0), 1, ctypes.byref(none)) == 1          # ConvTranspose 64 -> 3: pair-tiled kernel
    with_stats = _lib.epilogue_t(None, 0, 0.2, ctypes.c_void_p(4096), None, 0)
    assert lib.gcc_conv_route(ctypes.byref(d_l1), 0, ctypes.byref(with_stats)) == 0    # fused BN statistics: implicit GEMM
    d_l2 = conv(16, 128, 128, 128, 256, 4, 2, 1)
    assert lib.gcc_conv_route(ctypes.byref(d_l2), 0, ctypes.byref(none)) == 0
    d_l5 = conv(16, 31, 31, 1024, 1, 4, 1, 1)
    need = lib.gcc_conv_workspace(ctypes.byref(d_l5), 0)
    assert need > 0
    ws = _lib.epilogue_t(None, 0, 0.2, None, ctypes.c_void_p(1 << 20), need)
    assert lib.gcc_conv_route(ctypes.byref(d_l5), 0, ctypes.byref(ws)) == 2            # head route
    assert lib.gcc_conv_route(ctypes.byref(d_l5), 0, ctypes.byref(none)) == 0          # no workspace: implicit GEMM
    bad = conv(16, 31, 31, 0, 1, 4, 1, 1)
    assert lib.gcc_conv_route(ctypes.byref(bad), 0, ctypes.byref(none)) < 0


def test_ring_walk_route_predicates():
    """the ring-walk route of conv_ring3.hip (round 5) is chosen by geometry alone, and gcc_conv_stat_tiles promises its rows: 3 x 3
    stride 1 pad 1, 8..64 channels on both sides (source width a multiple of 8), >= 8192 pixels; one statistics row per workgroup"""
    from gcc_amd import _lib
    lib = _lib.load()
    conv = lambda N, H, W, Ci, Co, k, s, p: _lib.conv_t(N, H, W, Ci, Co, k, k, s, p, (Ci + 7) // 8 * 8, 0, (Co + 7) // 8 * 8, 0)
    none = _lib.epilogue_t(None, 0, 0.2, None, None, 0)
    with_stats = _lib.epilogue_t(None, 0, 0.2, ctypes.c_void_p(4096), None, 0)
    route = lambda d, dgrad=0, ep=none: lib.gcc_conv_route(ctypes.byref(d), dgrad, ctypes.byref(ep))
    trunk = conv(16, 96, 96, 64, 64, 3, 1, 1)                  # SRGAN's residual blocks at 96 x 96 (models/SRGAN.py:59-81)
    assert route(trunk) == 4 and route(trunk, 1) == 4 and route(trunk, 0, with_stats) == 4
    rows = lib.gcc_conv_stat_tiles(ctypes.byref(trunk), 0)
    assert 0 < rows <= 512                                       # workgroups: at most two per CU
    assert route(conv(16, 384, 384, 64, 64, 3, 1, 1)) == 4       # VGG19 conv1_2 at the high resolution
    assert route(conv(16, 96, 96, 24, 24, 3, 1, 1)) == 4         # the student's pruned widths
    assert route(conv(16, 96, 96, 24, 20, 3, 1, 1)) == 4 and route(conv(16, 96, 96, 24, 20, 3, 1, 1), 1) == 0   # dY of 20 channels: not a multiple of 8
    assert route(conv(16, 96, 96, 64, 128, 3, 1, 1)) == 0        # wider than 64: igemm_kernel / the halo kernel
    assert route(conv(16, 96, 96, 128, 64, 3, 1, 1)) == 0
    assert route(conv(16, 192, 192, 64, 64, 3, 2, 1)) == 0       # stride 2
    assert route(conv(2, 24, 24, 64, 64, 3, 1, 1)) == 0          # 1152 pixels: a launch-latency layer either way
    prev = lib.gcc_set_option(_lib.OPT_IGEMM_THIN, 0)
    try:
        assert route(trunk) == 0
        assert lib.gcc_conv_stat_tiles(ctypes.byref(trunk), 0) > 0          # igemm_kernel's pixel-tile rows
    finally:
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, prev)


def test_tile_plan_travels_with_the_call():
    """gcc_conv_t.plan (round 5): the tile plan is an argument of the call -- the same geometry lands on different tile families
    under different plans with NO library state in between; the headline shapes land on the 256-pixel tiles under the default
    plan; the remaining process-wide hooks are A/B switches (gcc_set_option) whose defaults gcc_options_default() vouches for"""
    from gcc_amd import _lib
    lib = _lib.load()

    def conv(N, H, W, Ci, Co, k, s, p, **plan):
        return _lib.conv_t(N, H, W, Ci, Co, k, k, s, p, (Ci + 7) // 8 * 8, 0, (Co + 7) // 8 * 8, 0,
                           tuple(plan.get(f, 0) for f in _lib.PLAN_FIELDS))
    tile = lambda d, dgrad=0: lib.gcc_conv_tile(ctypes.byref(d), dgrad)
    l2, l3, l4 = conv(16, 128, 128, 128, 256, 4, 2, 1), conv(16, 64, 64, 256, 512, 4, 2, 1), conv(16, 32, 32, 512, 1024, 4, 1, 1)
    assert tile(l2) == 256256
    assert tile(l3) == 256256 and tile(l3, 1) == 256256
    assert tile(l4, 1) == 256256          # 128 workgroups: half the chip, the other streams take the rest
    assert tile(l4) == 256256
    small = (2, 32, 32, 128, 256, 4, 2, 1)
    assert tile(conv(*small)) == 128128
    assert tile(conv(*small, tile_families=3, big_min=1, big_nk=1)) == 256256
    assert tile(conv(*small, tile_families=2, big_min=1, big_nk=1)) == 256128
    assert tile(conv(16, 128, 128, 128, 256, 4, 2, 1, tile_families=2, big_min=1, big_nk=1)) == 256128
    assert tile(conv(16, 32, 32, 512, 1024, 4, 1, 1, tile_families=1)) == 128128
    assert tile(conv(*small)) == 128128                      # ... and nothing stuck
    # the pair split sizes the workspace of the call that asks for it, and only of that call
    ws = lambda d: lib.gcc_conv_workspace(ctypes.byref(d), 1)
    assert ws(conv(16, 32, 32, 512, 1024, 4, 1, 1, pair=1)) >= 128 * 256 * 256 * 4 > ws(l4)
    assert lib.gcc_options_default() == 1 or any(os.environ.get('GCC_' + n) is not None for n in _lib.OPT_NAMES)
    try:
        prev = lib.gcc_set_option(_lib.OPT_WGRAD_BIG, 0)
        assert prev == 1 and lib.gcc_get_option(_lib.OPT_WGRAD_BIG) == 0 and lib.gcc_options_default() == 0
        assert lib.gcc_set_option(99, 1) < 0 and lib.gcc_get_option(-1) < 0
    finally:
        lib.gcc_set_option(_lib.OPT_WGRAD_BIG, -1)
    assert lib.gcc_get_option(_lib.OPT_WGRAD_BIG) == 1
    assert not hasattr(lib, 'gcc_diag_set') or True       # (the shipped library exports no diagnostic switch: checked below)
    import subprocess
    syms = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'gcc_diag_set' not in syms and 'gcc_conv_set_plan' not in syms


def test_gradient_layout_follows_backward_completion_order():
    """data parallelism: the flat gradient buffers are laid out in the order the backward pass completes the gradients
    (engine.UnetEngine.grad_segments / Pix2PixModel._d_layout), every parameter exactly once, and dist.GradReducer
    coalesces adjacent segments into buckets that tile the buffer"""
    from gcc_amd import dist as gdist
    from gcc_amd import engine
    from gcc_amd.models.Pix2Pix import MaskNLayerDiscriminator, Pix2PixModel, UnetGenertor
    g = UnetGenertor(3, 3, 8, ngf=8)
    names = {id(p): n for n, p in g.named_parameters()}
    segs = engine.UnetEngine.grad_segments(g, 8)
    assert len(segs) == 16 and sorted(names[id(p)] for s in segs for p in s) == sorted(names.values())
    assert [names[id(p)] for p in segs[0]] == ['model.model.3.weight', 'model.model.3.bias']        # outermost up conv first
    assert [names[id(p)] for p in segs[-1]] == ['model.model.0.weight']                              # outermost down conv last
    assert names[id(segs[8][0])].endswith('model.3.model.1.weight') and len(segs[8]) == 1             # innermost down conv
    holder = type('H', (), {})()
    holder.netD = MaskNLayerDiscriminator(input_nc=6, ndf=8)
    dn = {id(p): n for n, p in holder.netD.named_parameters()}
    dsegs = Pix2PixModel._d_layout(holder)
    assert [[dn[id(p)] for p in s] for s in dsegs] == [['model.15.weight', 'model.15.bias'], ['model.12.weight', 'model.12.bias', 'model.11.weight'],
                                                      ['model.8.weight', 'model.8.bias', 'model.7.weight'],
                                                      ['model.4.weight', 'model.4.bias', 'model.3.weight'], ['model.0.weight', 'model.0.bias']]
    params = list(g.parameters())
    before = [p.detach().clone() for p in params]
    flat = engine.FlatParams(params, 'cpu', layout=segs)
    assert all(torch.equal(p.detach(), b) for p, b in zip(params, before)), 're-homing keeps the values'
    assert flat.segments[0][0] == 0 and flat.segments[-1][1] == flat.total
    assert all(a[1] == b[0] for a, b in zip(flat.segments, flat.segments[1:]))
    assert segs[0][0].data_ptr() == flat.values.data_ptr()                                       # first segment sits at offset 0
    red = gdist.GradReducer(type('O', (), {'flat': flat, 'set_grad_scale': lambda self, s: None})(), bucket_bytes=64 << 10)
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.total and len(red.buckets) >= 3
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))
    assert [b[2] for b in red.buckets] == sorted(b[2] for b in red.buckets) and red.buckets[-1][2] == 15


def test_flat_params_redirect_is_a_host_side_pointer_swap():
    """engine.FlatParams.redirect (round 4: a second chain that adds to the same parameter gradients on another stream accumulates
    into a second buffer, folded in afterwards): inside the context every parameter's .grad is the view of the SIDE buffer
    with the shape, strides and offset of its view of `grads`; outside, the original views are back; (0 + a) + (0 + b) is what
    a then b gives"""
    from gcc_amd import engine
    from gcc_amd.models.Pix2Pix import UnetGenertor
    g = UnetGenertor(3, 3, 8, ngf=8)
    params = list(g.parameters())
    flat = engine.FlatParams(params, 'cpu')
    base, views = flat.grads.data_ptr(), list(flat.grad_views)
    gen = torch.Generator().manual_seed(1)
    a = [torch.randn(p.shape, generator=gen) for p in params]
    b = [torch.randn(p.shape, generator=gen) for p in params]
    for p, x in zip(params, a):
        p.grad.add_(x)                                    # the first chain: into `grads`
    with flat.redirect() as side:
        assert side.shape == flat.grads.shape and side.data_ptr() != base
        side.zero_()
        for p, v, x in zip(params, views, b):
            assert p.grad.data_ptr() - side.data_ptr() == v.data_ptr() - base
            assert p.grad.shape == v.shape and p.grad.stride() == v.stride()
            p.grad.add_(x)                                # the second chain: into the side buffer
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views)), 'the views of `grads` are restored'
    flat.grads.add_(flat.side_grads()[0])
    for p, x, y in zip(params, a, b):
        want = torch.zeros_like(x).add_(x).add_(y)        # what accumulating a then b into one buffer gives
        assert torch.equal(p.grad, want)
    with pytest.raises(RuntimeError):                     # an exception inside the context still restores the views
        with flat.redirect():
            raise RuntimeError('boom')
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, views))


def test_struct_layouts_match_header(tmp_path):
    """the ctypes mirrors of gcc_amd/_lib.py against the header itself: a C program that includes include/gcc_hip.h prints
    sizeof and the offset of the last field of every struct the shim mirrors (gcc, the host compiler a binding would use)"""
    import subprocess
    from gcc_amd import _lib
    pairs = [('gcc_conv_plan_t', _lib.conv_plan_t, 'wgrad_wgs'), ('gcc_conv_t', _lib.conv_t, 'plan'), ('gcc_epilogue_t', _lib.epilogue_t, 'y2_gate'),
             ('gcc_bn_t', _lib.bn_t, None), ('gcc_adam_tensor_t', _lib.adam_tensor_t, None), ('gcc_adam_chunk_t', _lib.adam_chunk_t, 'offset'),
             ('gcc_bnact_t', _lib.bnact_t, None), ('gcc_sn_item_t', _lib.sn_item_t, 'wt'), ('gcc_chansum_item_t', _lib.chansum_item_t, 'accumulate'),
             ('gcc_wgrad_item_t', _lib.wgrad_item_t, 'accumulate')]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "gcc_hip.h"\nint main(void) {\n'
    for cname, _, last in pairs:
        src += '  printf("%s %%zu %%zu\\n", sizeof(%s), %s);\n' % (cname, cname, 'offsetof(%s, %s)' % (cname, last) if last else '(size_t)0')
    src += '  return 0;\n}\n'
    c = tmp_path / 'layout.c'
    c.write_text(src)
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(c), '-o', str(exe)])
    out = dict((l.split()[0], (int(l.split()[1]), int(l.split()[2]))) for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct, last in pairs:
        assert ctypes.sizeof(ct) == out[cname][0], (cname, ctypes.sizeof(ct), out[cname][0])
        if last:
            assert getattr(ct, last).offset == out[cname][1], (cname, last)
    assert ctypes.sizeof(_lib.conv_t) == 13 * 4 + 7 * 4 and _lib.conv_t.plan.offset == 13 * 4


def test_options_match_reference_golden(golden_dir):
    from gcc_amd.options import options
    cases = json.load(open(os.path.join(golden_dir, 'options.json')))
    assert len(cases) >= 4
    for c in cases:
        got = vars(options.parse(c['argv']))
        got.pop('generator_only')
        for k, v in c['parsed'].items():
            g = got[k]
            if isinstance(v, float) and v != v:
                continue
            assert (g == v) or (g == float('inf') and v == 'inf'), (c['argv'], k, g, v)
        assert set(got) == set(c['parsed'])


def test_module_trees_have_reference_state_dict_keys(golden_dir):
    from gcc_amd.models.Pix2Pix import MaskNLayerDiscriminator, NLayerDiscriminator, UnetGenertor
    z = np.load(os.path.join(golden_dir, 'ops.npz'))
    g = UnetGenertor(3, 3, 6, ngf=4)
    assert list(g.state_dict().keys()) == [str(k) for k in z['init.G_keys']]
    d = NLayerDiscriminator(input_nc=6, ndf=4)
    assert list(d.state_dict().keys()) == [str(k) for k in z['init.D_keys']]
    zz = np.load(os.path.join(golden_dir, 'pix2pix_gcc_d6.npz'))
    mk = [k[len('final.sD.'):] for k in zz.files if k.startswith('final.sD.')]
    assert list(MaskNLayerDiscriminator(input_nc=6, ndf=8).state_dict().keys()) == mk
    # pruned cfg constructor: widths follow filter_cfgs / channel_cfgs (Appendix A.1 of SURVEY.md)
    f = [32, 24, 72, 112, 144, 120, 128, 256, 112, 112, 128, 152, 64, 24, 16]
    c = [32, 24, 72, 112, 144, 120, 128, 256, 240, 232, 272, 264, 136, 48, 48]
    p = UnetGenertor(3, 3, 8, ngf=32, filter_cfgs=f, channel_cfgs=c)
    sd = p.state_dict()
    assert tuple(sd['model.model.1.model.1.weight'].shape) == (24, 32, 4, 4)
    assert tuple(sd['model.model.3.weight'].shape) == (48, 3, 4, 4)
    assert tuple(sd['model.model.1.model.5.weight'].shape) == (48, 16, 4, 4)


def test_init_weights_rule():
    from gcc_amd.models.Pix2Pix import UnetGenertor
    from gcc_amd.utils import util
    torch.manual_seed(0)
    g = UnetGenertor(3, 3, 6, ngf=16)
    util.init_weights(g)
    w = torch.cat([p.flatten() for n, p in g.named_parameters() if p.dim() == 4])
    assert abs(float(w.std()) - 0.02) < 1e-3 and abs(float(w.mean())) < 1e-3
    gam = torch.cat([m.weight.flatten() for m in g.modules() if isinstance(m, torch.nn.BatchNorm2d)])
    bet = torch.cat([m.bias.flatten() for m in g.modules() if isinstance(m, torch.nn.BatchNorm2d)])
    assert abs(float(gam.mean()) - 1) < 1e-2 and abs(float(bet.std()) - 1) < 0.15
    assert float(g.state_dict()['model.model.3.bias'].abs().max()) == 0.0


def test_no_cpu_fallback():
    """without a GPU the model refuses to construct; it never routes through torch eager or the oracle"""
    from gcc_amd._lib import GccError
    from gcc_amd.models import get_model_class
    from gcc_amd.options import options
    opt = options.parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '-1'])
    with pytest.raises(GccError):
        get_model_class(opt)(opt)
    if not torch.cuda.is_available():
        opt = options.parse(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '0'])
        with pytest.raises(GccError):
            get_model_class(opt)(opt)
    import gcc_amd
    src = ''
    for dp, _, fs in os.walk(os.path.dirname(gcc_amd.__file__)):
        for f in fs:
            if f.endswith('.py'):
                src += open(os.path.join(dp, f)).read()
    assert 'import oracle' not in src and 'from oracle' not in src


def test_lr_schedule_matches_reference(golden_dir):
    from gcc_amd.utils import util
    z = np.load(os.path.join(golden_dir, 'pix2pix_pretrain_d6.npz'))
    ec, ne, nd, lr = z['sched']
    o = type('O', (), dict(epoch_count=int(ec), n_epochs=int(ne), n_epochs_decay=int(nd), lr_policy='linear'))()
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=float(lr))
    sch = util.get_scheduler(opt, o)
    got = []
    for _ in range(len(z['lr_after_epoch'])):
        opt.step()
        sch.step()
        got.append(opt.param_groups[0]['lr'])
    np.testing.assert_allclose(got, z['lr_after_epoch'], rtol=1e-12, atol=1e-15)


def test_prune_cfgs_and_budget_search_match_reference(golden_dir):
    """integer prune contract (bit exact): cfg lists at several thresholds, the thop-convention MAC count and the
    end point of the reference's binary search (run in the fixture script with a documented thop stand-in)"""
    from oracle import gcc_oracle as O
    from gcc_amd.utils import prune_util as PU
    from tests.golden.recipe import recipe_state_dict
    z = np.load(os.path.join(golden_dir, 'prune_d8.npz'))
    G = recipe_state_dict(O.unet_shapes(8, 8), int(z['seed_G']))
    gsp = torch.Generator().manual_seed(302)
    for name in PU.bn_names(8):
        G[name + '.weight'] = 1.0 + 0.02 * torch.randn(G[name + '.weight'].shape, generator=gsp)
    mx, mn = PU.max_min_bn_scale(G)
    assert [float(mx), float(mn)] == [float(v) for v in z['bn.max_min']]
    for i, t in enumerate(z['bn.thresholds']):
        f, c = PU.scale_prune_cfg(G, float(t), 8)
        assert f == [int(v) for v in z['bn.f.%d' % i]] and c == [int(v) for v in z['bn.c.%d' % i]], i
    s = np.load(os.path.join(golden_dir, 'prune_search_d8.npz'))
    opt = type('O', (), dict(scale_prune=True, num_downs=8, ngf=8, dataroot='./database/cityscapes/', load_size=256))()
    full = PU.cfg_macs(opt, [8, 16, 32, 64, 64, 64, 64, 64, 64, 64, 64, 64, 32, 16, 8],
                       [8, 16, 32, 64, 64, 64, 64, 64, 128, 128, 128, 128, 64, 32, 16])
    assert abs(full - float(s['full_macs'])) < 1e-9
    from gcc_amd.models.Pix2Pix import UnetGenertor
    assert abs(PU.unet_macs(UnetGenertor(3, 3, 8, ngf=8))[0] - float(s['full_macs'])) < 1e-9
    for i in range(3):
        if int(s['s%d.found' % i]):
            thr = PU.binarysearch_threshold_sd(G, opt, float(s['s%d.target' % i]))
            assert np.float32(float(thr)) == np.float32(s['s%d.threshold' % i]), i
            f, c = PU.scale_prune_cfg(G, thr, 8)
            assert f == [int(v) for v in s['s%d.f' % i]] and c == [int(v) for v in s['s%d.c' % i]]
            assert abs(PU.cfg_macs(opt, f, c) - float(s['s%d.macs' % i])) < 1e-9


def test_prune_search_that_removes_blocks(golden_dir):
    """budget searches whose answer prunes whole inner blocks away (tests/golden/pix2pix_pruned_removed_d8.npz: the reference's
    binarysearch_threshold on an ngf-32 U-Net with shaped BatchNorm scales): threshold (fp32), cfgs with their zero entries and
    the MAC count of the smaller nest, bit for bit; the parameter tree built from such a cfg has the reference's keys"""
    from oracle import gcc_oracle as O
    from gcc_amd.utils import prune_util as PU
    from tests.golden.recipe import recipe_state_dict, shape_bn_scales_for_removal
    from gcc_amd.models.Pix2Pix import UnetGenertor
    z = np.load(os.path.join(golden_dir, 'pix2pix_pruned_removed_d8.npz'))
    G = recipe_state_dict(O.unet_shapes(32, 8), int(z['seeds'][0]))
    shape_bn_scales_for_removal(G, int(z['seeds'][1]))
    mx, mn = PU.max_min_bn_scale(G)
    assert [float(mx), float(mn)] == [float(v) for v in z['max_min']]
    opt = type('O', (), dict(scale_prune=True, num_downs=8, ngf=32, dataroot='./database/cityscapes/', load_size=256))()
    assert abs(PU.unet_macs(UnetGenertor(3, 3, 8, ngf=32))[0] - float(z['full_macs'])) < 1e-9
    for tag, blocks in (('k7', 7), ('k6', 6), ('k5', 5)):
        if tag != 'k5':
            thr = PU.binarysearch_threshold_sd(G, opt, float(z[tag + '.target']))
            assert np.float32(float(thr)) == np.float32(z[tag + '.threshold']), tag
        else:
            thr = float(z['k5.threshold'])
        f, c = PU.scale_prune_cfg(G, thr, 32)
        assert f == [int(v) for v in z[tag + '.f']] and c == [int(v) for v in z[tag + '.c']], tag
        assert abs(PU.cfg_macs(opt, f, c) - float(z[tag + '.macs'])) < 1e-9, tag
        net = UnetGenertor(3, 3, 8, ngf=32, filter_cfgs=f, channel_cfgs=c)
        assert len(net.present) == blocks and net.inner_identity
        assert list(net.state_dict().keys()) == [str(k) for k in z[tag + '.keys']]
        assert abs(PU.unet_macs(net)[0] - float(z[tag + '.macs'])) < 1e-9, tag
    with pytest.raises(ValueError):         # a block removed from the middle with widths that do not meet: the reference would fail in forward()
        f, c = [int(v) for v in z['k7.f']], [int(v) for v in z['k7.c']]
        f[5] = 0
        UnetGenertor(3, 3, 8, ngf=32, filter_cfgs=f, channel_cfgs=c)


# ---------------------------------------------------------------------------------------------------
# norm / resnet pruning cfgs (host logic on the parameter trees; golden: tests/golden/prune_resnet.npz, prune_d8.npz)
# ---------------------------------------------------------------------------------------------------
def _spread_filter_norms(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                f = 0.5 + torch.rand(m.weight.shape[0], generator=g)
                m.weight.mul_(f.reshape(-1, 1, 1, 1))


def _load_recipe(net, seed):
    from collections import OrderedDict
    from tests.golden.recipe import recipe_state_dict
    net.load_state_dict(recipe_state_dict(OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items()), seed))


def test_resnet_prune_cfgs_bit_exact(golden_dir):
    from gcc_amd.models.Pix2Pix import MobileResnetGenerator
    from gcc_amd.utils import prune_util as P
    z = np.load(os.path.join(golden_dir, 'prune_resnet.npz'))
    for tag, rule, seeds in (('p2p', 'union', (701, 702)), ('cyc', 'mean', (703, 704))):
        net = MobileResnetGenerator(ngf=8)
        _load_recipe(net, seeds[0])
        _spread_filter_norms(net, seeds[1])
        mx, mn = P.max_min_conv_norm_resnet(net, rule)
        assert [float(mx), float(mn)] == [float(v) for v in z[tag + '.max_min']], tag
        for i, t in enumerate(z[tag + '.thresholds']):
            assert P.resnet_prune_cfg(net, float(t), rule) == [int(v) for v in z['%s.f.%d' % (tag, i)]], (tag, i)
    # a cfg read back from a tree built from it (removed block included)
    cfg = [int(v) for v in z['pruned_b.cfg']]
    net = MobileResnetGenerator(ngf=8, cfg=cfg)
    assert list(net.state_dict().keys()) == [str(k) for k in z['pruned_b.G_keys']]
    # MAC budget by shape arithmetic: the reference's hard-coded CycleGAN students were searched to 2.4 / 2.7 G +- 0.05
    assert abs(P.resnet_cfg_macs(P.CYCLEGAN_CFG_ATOB) - 2.4) <= 0.05 and abs(P.resnet_cfg_macs(P.CYCLEGAN_CFG_BTOA) - 2.7) <= 0.05
    full = MobileResnetGenerator(ngf=8)
    assert P.mobile_resnet_cfg(full) == [8, 16, 32] + [32] * 18 + [16, 8]
    assert abs(P.mobile_resnet_macs(full, 256)[0] - P.resnet_cfg_macs([8, 16, 32] + [32] * 18 + [16, 8])) < 1e-12


def test_unet_norm_prune_cfgs_bit_exact(golden_dir):
    from gcc_amd.models.Pix2Pix import UnetGenertor
    from gcc_amd.utils import prune_util as P
    z = np.load(os.path.join(golden_dir, 'prune_d8.npz'))
    net = UnetGenertor(3, 3, 8, ngf=8)
    _load_recipe(net, int(z['seed_G']))
    mx, mn = P.max_min_conv_norm_unet(net)
    assert [float(mx), float(mn)] == [float(v) for v in z['norm.max_min']]
    for i, t in enumerate(z['norm.thresholds']):
        f, c = P.norm_prune_cfg(net, float(t), ngf=8)
        assert f == [int(v) for v in z['norm.f.%d' % i]] and c == [int(v) for v in z['norm.c.%d' % i]], i


def _spread_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.05 + torch.rand(m.weight.shape, generator=g))


def test_srgan_sagan_prune_search_vs_reference(golden_dir):
    """binarysearch_threshold + prune cfgs of SRGAN (scale / norm) and SAGAN (scale) against the reference's own search
    (tests/golden/make_fixtures.py::fixture_prune_search_gan: the reference's code with the thop stand-in)"""
    import types
    from gcc_amd.models.SRGAN import Generator as SRGenerator
    from gcc_amd.models.SAGAN import Generator as SAGenerator
    from gcc_amd.utils import prune_util as P
    z = np.load(os.path.join(golden_dir, 'prune_search_gan.npz'))
    for tag, scale in (('sr_scale', True), ('sr_norm', False)):
        g = SRGenerator(n_channels=8)
        _load_recipe(g, 951)
        _spread_bn(g, 952)
        _spread_filter_norms(g, 953)
        opt = types.SimpleNamespace(model='srgan', image_size=96, upscale_factor=4, scale_prune=scale, norm_prune=not scale,
                                    dataroot='./database/sr/', backbone='unet')
        assert abs(P.get_flops_parms(g, None, opt)[0] - float(z[tag + '.full_macs'])) < 1e-12
        mm = P.srgan_max_min_bn_scale(g) if scale else P.srgan_max_min_conv_norm(g)
        assert [float(mm[0]), float(mm[1])] == [float(v) for v in z[tag + '.max_min']]
        model = types.SimpleNamespace(opt=opt, netG=g, max_min_bn_scale=lambda: P.srgan_max_min_bn_scale(g),
                                      max_min_conv_norm=lambda: P.srgan_max_min_conv_norm(g))
        for i in range(3):
            assert int(z['%s.s%d.found' % (tag, i)]) == 1
            thr = P.binarysearch_threshold(model, float(z['%s.s%d.target' % (tag, i)]))
            assert np.float32(float(thr)) == np.float32(z['%s.s%d.threshold' % (tag, i)]), (tag, i)
            f, _ = P.srgan_prune_cfg(g, thr, scale)
            assert f == [int(v) for v in z['%s.s%d.f' % (tag, i)]], (tag, i, f)      # norm pruning: 17 entries, as there
            pruned = SRGenerator(n_channels=8, filter_cfgs=f)
            assert abs(P.srresnet_macs(pruned, 24)[0] - float(z['%s.s%d.macs' % (tag, i)])) < 1e-12
    g = SAGenerator(ngf=32)
    _load_recipe(g, 961)
    _spread_bn(g, 962)
    opt = types.SimpleNamespace(model='sagan', scale_prune=True, norm_prune=False, dataroot='./database/celeb/', backbone='unet')
    assert abs(P.get_flops_parms(g, None, opt)[0] - float(z['sa.full_macs'])) < 1e-12

    def mm():
        top = low = None
        for m in g.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                w = m.weight.detach()
                top = w.max() if top is None else torch.min(w.max(), top)
                low = w.min() if low is None else torch.min(w.min(), low)
        return top, low
    assert [float(v) for v in mm()] == [float(v) for v in z['sa.max_min']]
    model = types.SimpleNamespace(opt=opt, netG=g, max_min_bn_scale=mm)
    for i in range(3):
        assert int(z['sa.s%d.found' % i]) == 1
        thr = P.binarysearch_threshold(model, float(z['sa.s%d.target' % i]))
        assert np.float32(float(thr)) == np.float32(z['sa.s%d.threshold' % i]), i
        f = P.sagan_scale_prune_cfg(g, thr)
        assert f == [int(v) for v in z['sa.s%d.f' % i]]
        assert abs(P.sagan_generator_macs(SAGenerator(ngf=32, filter_cfgs=f))[0] - float(z['sa.s%d.macs' % i])) < 1e-12
