#!/usr/bin/env python3
"""Headline benchmark: training images/s of the Pix2Pix GCC iteration at 256x256 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one synthetic batch of 16 paired 256x256 images per GPU:
    set_input(train) -> optimize_parameters() -> set_input(val) -> clipping_mask_alpha() ->
    optimizer_netD_arch()                                   (reference train.py:144-151)
for BASELINE.json configs[1]: student U-Net ngf 32 + selective-activation PatchGAN ndf 128, online
teacher ngf 64 / ndf 128, hinge GAN loss, lambda_L1 100, content 50, gram 1e4, bf16 MFMA compute.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.

roofline: the dominant kernel family is the implicit-GEMM convolution (igemm_kernel / igemm_halo_kernel: conv fprop /
dgrad / ConvTranspose).  Every launch of it inside the first step of the timed region is bracketed by HIP
events on the launch stream (bracketing all steps costs ~6% throughput); achieved = sum of algorithmic FLOPs
(2*M*Cout*taps*Cin, padding excluded) / sum of measured durations; peak = 2.5 PFLOP/s dense bf16 MFMA
(MI355X_MICROARCH.md).  That step runs with the production streams (student, online teacher, auxiliary, weight
gradients) folded onto one, so that a launch's duration is the kernel's own and not its neighbours' share of the CUs,
under the tile plan the production schedule uses (`roofline.frac`); one more such step AFTER the timed region runs the
plan for launches that have the chip to themselves (pair split, full-chip weight-gradient splits:
`roofline.frac_alone_plan`).  The other
steps of the timed region run the production schedule (profiles/: rocprofv3 summaries of `bench.py
--serialize-streams`, which agree with these durations, and of the default command).
other_configs: after the timed region and outside `value`, one short run each of BASELINE.json configs 3-5 (CycleGAN,
SAGAN, SRGAN at their reference widths, synthetic batches, 10 warm-up + 30 timed iterations).
cpu_baseline: the oracle (CPU restatement pinned to the reference) timed on this host's cores on a
bounded sample (N=1, same architecture, 1 warm-up + 16 timed iterations, about 10 s), rank 0 at --gpus 1 only.
"""
import argparse
import copy
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # this pool's driver supports dmabuf IPC only (RCCL needs it)
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GCC_ARGV = ['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--ngf', '32', '--ndf', '128',
            '--online_distillation', '--darts_discriminator', '--lambda_content', '50', '--lambda_gram', '1e4',
            '--arch_lr', '1e-4', '--arch_lr_step', '--gpu_ids', '0']
FLOP_PER_IMG = 679.3e9      # 514.0 (optimize_parameters) + 165.3 (arch step) GFLOP
PEAK_BF16 = 2.5e15


def log(msg):
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %s] %s' % (time.strftime('%H:%M:%S'), msg), file=sys.stderr, flush=True)


def build(batch):
    from gcc_amd.options import options
    from gcc_amd.models import get_model_class
    opt = options.parse(GCC_ARGV)
    opt.isTrain = True
    opt.batch_size = batch
    cls = get_model_class(opt)
    torch.manual_seed(0)
    model = cls(opt)
    topt = copy.deepcopy(opt)
    topt.ngf, topt.ndf = opt.teacher_ngf, opt.teacher_ndf
    topt.darts_discriminator = topt.online_distillation = False
    teacher = cls(topt)
    teacher.model_train()
    model.teacher_model = teacher
    model.init_distillation()
    teacher.init_distillation()
    model.model_train()
    return model, opt


def synthetic(batch, rank, device, size=256):
    def pair(seed):
        g = torch.Generator().manual_seed(seed)
        a = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
        b = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
        d = {'A': a.to(device), 'B': b.to(device), 'A_paths': [''] * batch, 'B_paths': [''] * batch}
        # resident in HBM before the timed region: the 'ready' event (gcc_amd.models._streams: what a prefetching loader hands
        # over with a device batch) tells set_input that no producer kernel is pending
        torch.cuda.synchronize()
        d['ready'] = torch.cuda.Event()
        d['ready'].record()
        return d
    return pair(1234 + rank), pair(4321 + rank)


def serialize_streams(model, engine, flag, plan=None):
    """the production schedule runs four HIP streams (student, online teacher, auxiliary, weight gradients): kernels of
    different streams share the CUs, so a launch's wall duration there includes its neighbours'.  The roofline block wants
    the kernel's own duration: the profiled steps run with everything on one stream, under the tile plan named by `plan`
    (models/_streams.py set_stream_schedule: 'production' or 'alone')."""
    model.set_stream_schedule(not flag, plan)


def one_step(model, train, val):
    model.set_input(train)
    model.optimize_parameters()
    model.set_input(val)
    model.clipping_mask_alpha()
    model.optimizer_netD_arch()


def cpu_baseline(iters=16, batch=1, budget=30.0):
    """oracle on the host cores: `batch` images per iteration (N=1 and N=4), full-size networks; a bounded sample
    (N=1: 16 iterations at ~0.6 s, fewer if the host is slower; at most `budget` seconds)"""
    from oracle import gcc_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))     # measured on the 256-core host: 16 threads 0.62 s/it, 32: 0.89, 64: 79 (oversubscribed)
    torch.set_num_threads(ncores)
    opt = O.Opt(ngf=32, ndf=128, teacher_ngf=64, teacher_ndf=128, num_downs=8, no_dropout=False, direction='BtoA')
    m = O.build_gcc_pair(opt, seed=0)
    g = torch.Generator().manual_seed(1234)
    A, B, vA, vB = (torch.rand(batch, 3, 256, 256, generator=g) * 2 - 1 for _ in range(4))

    def it():
        m.set_input(A, B)
        m.optimize_parameters()
        m.set_input(vA, vB)
        m.clipping_mask_alpha()
        m.optimizer_netD_arch()
    t0 = time.time()
    it()
    first = time.time() - t0
    log('cpu_baseline: N=%d warm-up iteration %.1f s on %d threads' % (batch, first, ncores))
    iters = max(1, min(iters, int(budget / max(first, 1e-3))))
    t0 = time.time()
    for _ in range(iters):
        it()
    dt = (time.time() - t0) / iters
    return {'value': round(batch / dt, 4), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'N=%d 256x256, same GCC iteration (student ngf32/ndf128 + teacher ngf64/ndf128), fp32, '
                      '1 warm-up + %d timed iterations, %.2f s/iteration' % (batch, iters, dt)}


def allreduce_table(model, device, reps=5):
    """after the timed region: every gradient bucket of the step all-reduced alone (blocking, `reps` times, max over ranks)
    -- what the exchange costs when nothing overlaps it, so that a scaling curve can be read against it -- through
    torch.distributed (ProcessGroupNCCL's stream) and, with GCC_DP_COMM=native or GCC_BENCH_NATIVE_TABLE=1 on an RCCL backend,
    through the C ABI's communicator too (gcc_comm_allreduce_sum_f32 on the caller's stream)."""
    import torch.distributed as dist
    from gcc_amd import dist as gdist
    T = model.teacher_model
    groups = []
    for name, opt_ in (('teacher_D', T.optimizer_D), ('teacher_G', T.optimizer_G), ('student_D', model.optimizer_D),
                       ('student_G', model.optimizer_G), ('alpha', model.optimizer_arch)):
        red = getattr(opt_, 'reducer', None)
        spans = [(b, e) for b, e, _ in red.buckets] if red is not None else [(0, opt_.flat.grads.numel())]
        groups += [('%s[%d]' % (name, i), opt_, b, e) for i, (b, e) in enumerate(spans)]

    def table(reduce_fn):
        rows, total_ms, total_mb = [], 0.0, 0.0
        for label, opt_, b, e in groups:
            buf = opt_.flat.grads[b:e].clone()
            reduce_fn(buf)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                reduce_fn(buf)
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms, mb = float(t.item()), (e - b) * 4 / 1e6
            rows.append({'bucket': label, 'MB': round(mb, 2), 'ms': round(ms, 3), 'algbw_GBps': round(mb / ms, 1) if ms > 0 else None})
            total_ms += ms
            total_mb += mb
        return {'buckets': rows, 'total_MB_fp32': round(total_mb, 1), 'total_ms_unoverlapped': round(total_ms, 3)}
    out = {'route': gdist.comm_route(), 'bucketed': getattr(model.optimizer_D, 'reducer', None) is not None,
           'bucket_dtype_in_step': 'bf16 (GCC_DP_BF16=1)' if gdist.bf16_buckets() else 'fp32',
           'note': 'each bucket alone after the timed region; in the step they overlap the backward pass (dist.GradReducer)',
           'torch': table(lambda buf: dist.all_reduce(buf))}
    # the second table needs the C ABI's communicator: its creation is one more collective (ncclCommInitRank) that has never run
    # on two devices of this pool (the one-GPU test boxes skip that test), so it is taken only where the native route is the one
    # in use, or on request (GCC_BENCH_NATIVE_TABLE=1) -- a hang here would cost the whole scaling line
    if dist.get_backend() == 'nccl' and (gdist.comm_route() == 'native' or os.environ.get('GCC_BENCH_NATIVE_TABLE') == '1'):
        try:
            comm = gdist.native_comm()
            out['native'] = table(lambda buf: comm.all_reduce_sum_(buf))
        except Exception as e:
            out['native'] = {'error': '%s: %s' % (type(e).__name__, e)}
    return out


def loss_check_step1(first, batch, world, rel=2e-2):
    """The losses of the very first iteration (seeded weights, seeded synthetic batch, counter-RNG dropout: the same numbers on
    every run of an unchanged tree) against the committed values of tests/golden/bench_step1_losses.json -- a changed
    kernel that shifts the arithmetic shows up here, on the bench's own full-size workload.  World 1 only: with more
    ranks the architecture step of iteration 1 already runs on all-reduced weights."""
    if first is None:
        return None
    out = {'values': {k: round(v, 5) for k, v in first.items()}}
    path = os.path.join(ROOT, 'tests', 'golden', 'bench_step1_losses.json')
    if world != 1 or not os.path.exists(path):
        out['reference'] = None
        return out
    with open(path) as fh:
        ref = json.load(fh).get('batch%d' % batch)
    if ref is None:
        out['reference'] = None
        return out
    dev = {k: abs(first[k] - ref[k]) / max(abs(ref[k]), 1e-3) for k in ref if k in first}
    out.update({'kind': 'self: a regression guard against values this HIP path produced at an earlier tree, NOT parity -- parity at this '
                        'size, dropout on, is tests/test_pix2pix_gpu.py::test_full_config_dropout_iteration_vs_oracle (CPU oracle handed the kernels\' masks) '
                        'and ::test_full_config_gradients_vs_oracle[16] (every parameter gradient)',
                'reference': 'tests/golden/bench_step1_losses.json', 'max_rel_dev': round(max(dev.values()), 6),
                'tolerance': rel, 'ok': bool(max(dev.values()) <= rel and set(ref) <= set(first))})
    if not out['ok']:
        log('WARNING: first-iteration losses differ from the committed values: %s' % dev)
    return out


def generator_block(tag_stats, batch, spans=None):
    """north_star's sub-figure: MFMA utilisation of the generator forward + backward at batch 16.  From the bracketed step:
    conv_* = the generator's fprop / dgrad / wgrad launches alone (algorithmic FLOP / sum of their durations); pass_* = the
    same FLOP over the wall time of the whole G.forward + G.backward passes, BatchNorm / activation / dropout kernels
    included.  SURVEY 8(d): 9.25 (student) / 36.19 (teacher) GFLOP per image fwd + bwd; the step also holds the arch step's
    generator forward (3.10 / 12.10), counted in `measured`.  `spans` (a later step on the same single stream with only the
    passes timed, no launch bracketed): the pass_* figures then come from the launches the product path really runs -- the
    bracketed step replaces every fused conv + BatchNorm call by its separately timed parts and pays two events per launch
    -- and the bracketed step's wall time stays beside them as pass_ms_bracketed."""
    out = {}
    for who, gflop in (('student_G', 9.25), ('teacher_G', 36.19)):
        parts = [tag_stats.get(who + '.fwd'), tag_stats.get(who + '.bwd')]
        if not all(parts):
            continue
        flop = sum(p['flop'] for p in parts)
        conv_s = sum(p['conv_s'] for p in parts)
        span_s = sum(p.get('span_s', 0.0) for p in parts)
        brk = span_s
        if spans and (who + '.fwd') in spans and (who + '.bwd') in spans:
            span_s = spans[who + '.fwd'] + spans[who + '.bwd']
            for p, sfx in zip(parts, ('.fwd', '.bwd')):
                p['span_prod_s'] = spans[who + sfx]
        out[who] = {'survey_gflop_per_image_fwd_bwd': gflop, 'measured_gflop_per_image': round(flop / batch / 1e9, 2),
                    'conv_launches': sum(p['conv_launches'] for p in parts), 'conv_ms': round(conv_s * 1e3, 3),
                    'conv_tflops': round(flop / conv_s / 1e12, 1), 'conv_mfma_frac': round(flop / conv_s / PEAK_BF16, 4),
                    'pass_ms': round(span_s * 1e3, 3), 'pass_tflops': round(flop / span_s / 1e12, 1),
                    'pass_mfma_frac': round(flop / span_s / PEAK_BF16, 4), 'pass_ms_bracketed': round(brk * 1e3, 3),
                    'pass_timed_on': 'product path, passes only' if span_s != brk else 'bracketed step'}
    if len(out) == 2:
        tot = {k: sum(tag_stats[w + p][k] for w in ('student_G', 'teacher_G') for p in ('.fwd', '.bwd')) for k in ('flop', 'conv_s', 'span_s')}
        if spans and all('span_prod_s' in tag_stats[w + p] for w in ('student_G', 'teacher_G') for p in ('.fwd', '.bwd')):
            tot['span_s'] = sum(tag_stats[w + p]['span_prod_s'] for w in ('student_G', 'teacher_G') for p in ('.fwd', '.bwd'))
        out['both'] = {'conv_mfma_frac': round(tot['flop'] / tot['conv_s'] / PEAK_BF16, 4),
                       'pass_mfma_frac': round(tot['flop'] / tot['span_s'] / PEAK_BF16, 4), 'target': 0.40}
    return out


OTHER_ARGV = {
    'cyclegan': (1, ['--dataroot', 'synthetic', '--model', 'cyclegan', '--ngf', '24', '--ndf', '64', '--teacher_ngf', '64',
                     '--lambda_content', '0.01', '--lambda_gram', '10']),
    'sagan': (64, ['--dataroot', 'synthetic', '--model', 'sagan', '--ngf', '48', '--ndf', '64', '--teacher_ngf', '64',
                   '--crop_size', '64', '--gan_mode', 'hinge']),
    'srgan': (16, ['--dataroot', 'synthetic', '--model', 'srgan', '--ngf', '24', '--teacher_ngf', '64', '--image_size', '96']),
    # BASELINE.json config 5 at its literal size: x4 96 x 96 -> 384 x 384 (16 times the pixels of the training crop above)
    'srgan_96_to_384': (16, ['--dataroot', 'synthetic', '--model', 'srgan', '--ngf', '24', '--teacher_ngf', '64', '--image_size', '384']),
}


def other_configs(warmup=10, steps=30):
    """BASELINE.json configs 3-5 on synthetic batches at the reference scripts' widths (distillation + architecture step on):
    CycleGAN 256x256 batch 1, SAGAN 64x64 batch 64, SRGAN x4 24 -> 96 crops batch 16 (options/options.py:196-203).  Outside
    `value`; launch-bound on the eager host path, so launches per step are printed beside the time, and
    the replayed iteration's time beside both."""
    from gcc_amd import ops
    from gcc_amd.models import get_model_class
    from gcc_amd.options import options
    from gcc_amd.train import SyntheticPairs, attach_teacher
    os.environ.setdefault('GCC_VGG19_RANDOM', '1')       # torchvision's VGG19 weights cannot be downloaded here
    out = {}
    only = [w for w in os.environ.get('GCC_BENCH_OTHER', '').split(',') if w]        # tuning aid: a subset of the side configs
    for which, (batch, argv) in OTHER_ARGV.items():
        if only and which not in only:
            continue
        try:
            opt = options.parse(argv + ['--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--batch_size', str(batch)])
            opt.isTrain = True
            if getattr(opt, 'teacher_ndf', None) is None:
                opt.teacher_ndf = opt.ndf
            cls = get_model_class(opt)
            model = cls(opt)
            attach_teacher(model, opt, cls)
            model.model_train()
            # inputs resident in HBM before the timed region, as for the headline line
            data = [{k: (v.to(model.device) if torch.is_tensor(v) else v) for k, v in d.items()} for d in SyntheticPairs(opt, 4, 7)]

            def step(i):
                model.set_input(data[i % 4])
                model.optimize_parameters()
                model.set_input(data[(i + 1) % 4])
                model.clipping_mask_alpha()
                model.optimizer_netD_arch()
            for i in range(warmup):
                step(i)
            torch.cuda.synchronize()
            ops.lib().gcc_launch_count(1)
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            launches = ops.lib().gcc_launch_count(1) / steps
            out[which] = {'batch': batch, 'ms_per_step': round(ms, 3), 'images_per_s': round(batch / ms * 1e3, 1),
                          'launches_per_step': round(launches)}
            # roofline of the iteration: the FLOP of its MFMA conv launches (counted by one bracketed step on one stream)
            # over the eager and the replayed step time, against the dense bf16 peak; the conv roofline beside it
            try:
                torch.cuda.synchronize()
                ops.PROFILE.start(steps=1)
                step(0)
                ops.PROFILE.step_done()
                r = ops.PROFILE.stop()
                if r is not None:
                    fl = sum(v['gflop'] for v in r['per_kernel'].values()) * 1e9
                    out[which]['roofline'] = {
                        'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': PEAK_BF16 / 1e12,
                        'conv_gflop_per_step': round(fl / 1e9, 2),
                        'achieved': round(fl / (ms * 1e-3) / 1e12, 2), 'frac': round(fl / (ms * 1e-3) / PEAK_BF16, 4),
                        'dominant_kernel': r['kernel'], 'dominant_kernel_frac': r['frac'],
                        'conv_roofline': r.get('conv_roofline'), 'per_kernel': r['per_kernel']}
            except Exception as e:
                out[which]['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
            # the same iteration recorded once and re-issued from native code (gcc_amd.replay; bit-identical results:
            # tests/test_replay_gpu.py): the figure a launch-bound model trains at with GCC_REPLAY=1
            try:
                from gcc_amd.replay import IterationReplay
                rp = IterationReplay(model, opt, warmup=1, enabled=True)
                i = 0
                while rp.rec is None and i < 4:
                    rp.step(data[i % 4], data[(i + 1) % 4])
                    i += 1
                for j in range(5):
                    rp.step(data[(i + j) % 4], data[(i + j + 1) % 4])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for j in range(steps):
                    rp.step(data[j % 4], data[(j + 1) % 4])
                torch.cuda.synchronize()
                rms = (time.perf_counter() - t0) / steps * 1e3
                info = rp.info() or {}
                if 'conv_gflop_per_step' in out[which].get('roofline', {}):
                    out[which]['roofline']['frac_replayed'] = round(
                        out[which]['roofline']['conv_gflop_per_step'] * 1e9 / (rms * 1e-3) / PEAK_BF16, 4)
                out[which]['replay'] = {'ms_per_step': round(rms, 3), 'images_per_s': round(batch / rms * 1e3, 1),
                                        'entries': info.get('entries'), 'host_threads': info.get('threads'),
                                        'streams': info.get('streams')}
                rp.invalidate()
                del rp
            except Exception as e:
                out[which]['replay'] = {'error': '%s: %s' % (type(e).__name__, e)}
            del model, data
            torch.cuda.empty_cache()
        except Exception as e:      # the headline line must not be lost to a side measurement
            out[which] = {'error': '%s: %s' % (type(e).__name__, e)}
        log('other_configs %s: %s' % (which, out[which]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--serialize-streams', action='store_true',
                    help='run every step on one HIP stream (as the profiled steps do): for rocprofv3 runs whose per-kernel '
                         'average must be the kernel\'s own duration')
    args = ap.parse_args()

    from gcc_amd import dist as gdist
    from gcc_amd import ops
    world = gdist.init_from_env()
    rank = gdist.rank()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    # the measured path is the shipped library with every process-wide tuning hook at its built-in default (the tile plan is not
    # among them: it travels with each call, ops.set_plan): a GCC_* override in the environment would make the line describe
    # another configuration than the product's -- refuse (GCC_BENCH_ALLOW_OPTIONS=1: same-box A/B runs, flagged in the JSON)
    from gcc_amd import _lib
    lib = ops.lib()
    option_vector = {n: int(lib.gcc_get_option(i)) for i, n in enumerate(_lib.OPT_NAMES)}
    options_default = bool(lib.gcc_options_default())
    if 'diag' in os.path.basename(_lib.LIB_PATH):
        raise SystemExit('bench.py measures libgcc_hip.so, not the diagnostic build (%s)' % _lib.LIB_PATH)
    if not options_default and os.environ.get('GCC_BENCH_ALLOW_OPTIONS') != '1':
        raise SystemExit('tuning hooks differ from their defaults (%s): unset the GCC_* variables or pass GCC_BENCH_ALLOW_OPTIONS=1'
                         % {k: v for k, v in option_vector.items()})
    model, opt = build(args.batch)
    model.G.profile_tag, model.teacher_model.G.profile_tag = 'student_G', 'teacher_G'
    device = model.device
    torch.cuda.set_device(device)
    train, val = synthetic(args.batch, rank, device)

    log('model built on %s (world %d); warm-up %d steps' % (device, world, args.warmup))
    first_losses = None
    for i in range(args.warmup):
        one_step(model, train, val)
        if i == 0:
            torch.cuda.synchronize()
            log('first step done')
            first_losses = dict(model.get_current_losses())
    torch.cuda.synchronize()
    log('warm-up done; timing %d steps' % args.steps)
    if world > 1:
        torch.distributed.barrier()
    from gcc_amd import engine
    n_prof = 0 if args.no_roofline else min(args.steps, 1)
    if args.serialize_streams:
        serialize_streams(model, engine, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roofs = []
    gen_tags = None
    launches_per_step = None
    for i in range(args.steps):
        if i < n_prof:
            # profiled step: one stream, a launch's duration is its own; the tile plan is the production schedule's
            torch.cuda.synchronize()
            serialize_streams(model, engine, True, 'production')
            ops.PROFILE.start(steps=1)          # HIP events bracket the MFMA-kernel launches of this step
        elif i == n_prof and n_prof:
            torch.cuda.synchronize()
            serialize_streams(model, engine, args.serialize_streams)
        if i == n_prof:
            ops.lib().gcc_launch_count(1)
        one_step(model, train, val)
        if i == n_prof:
            launches_per_step = int(ops.lib().gcc_launch_count(0))
        if i < n_prof:
            ops.PROFILE.step_done()
            r = ops.PROFILE.stop()
            if i == 0 and r is not None:
                gen_tags = ops.PROFILE.tag_stats
            roofs.append(r)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    log('timed region: %.3f s (%.1f ms/step)' % (dt, 1000 * dt / args.steps))
    if n_prof:
        # after the timed region: one more single-stream step under the plan for launches that have the chip to themselves
        serialize_streams(model, engine, True, 'alone')
        one_step(model, train, val)                  # the split plan changes workspace sizes: one untimed step first
        ops.PROFILE.start(steps=1)
        one_step(model, train, val)
        ops.PROFILE.step_done()
        roofs.append(ops.PROFILE.stop())
        # and one under the production plan with only the generator passes timed (no bracketed launch, every fused route on)
        serialize_streams(model, engine, True, 'production')
        one_step(model, train, val)
        ops.PROFILE.start_spans()
        one_step(model, train, val)
        spans = ops.PROFILE.stop_spans()
        if roofs[0] is not None and gen_tags is not None:
            roofs[0]['generator'] = generator_block(gen_tags, args.batch, spans)
        serialize_streams(model, engine, args.serialize_streams)
    roof = None
    if roofs and roofs[0] is not None:
        roof = roofs[0]
        roof['plan'] = 'production tile plan (no pair split, half-chip weight-gradient splits), one stream'
        if len(roofs) > 1 and roofs[1] is not None:
            alone = roofs[1]
            roof['frac_alone_plan'] = alone['frac']
            roof['achieved_alone_plan'] = alone['achieved']
            roof['alone_plan'] = {'plan': 'pair split of half-chip 256x256 launches, full-chip weight-gradient splits, one stream',
                                  'per_kernel': alone['per_kernel'], 'conv_roofline_frac': alone.get('conv_roofline', {}).get('frac')}
        roof['steps_bracketed'] = 1
        # HBM bytes per launch of the igemm family from the PMC counters: collected in their own rocprofv3 passes (--pmc
        # FETCH_SIZE, --pmc WRITE_SIZE; scratch/pmc_traffic.py applies the gfx950 corrections) and committed under profiles/
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_igemm_hbm_traffic.json')))
        if files:
            with open(files[-1]) as fh:
                tj = json.load(fh)
            roof['traffic'] = round(tj['hbm_bytes_per_launch_corrected'])
            roof['traffic_source'] = 'profiles/%s (bytes per launch, rocprofv3 PMC passes of tree %s)' % (
                os.path.basename(files[-1]), tj.get('git_head', 'of round 2'))
            if 'per_family' in tj:
                roof['traffic_per_family'] = tj['per_family']
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.get_current_losses()
    comm = allreduce_table(model, device) if world > 1 else None
    from gcc_amd import dist as gdist
    rccl_ranks = gdist.rccl_ranks() if world > 1 else 1          # collective: every rank calls it
    if rank != 0:
        return
    imgs = world * args.batch * args.steps
    value = imgs / dt
    out = {
        'metric': 'training images/sec at 256x256 (pix2pix GCC)', 'value': round(value, 2), 'unit': 'images/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1000.0 * dt / args.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'Pix2Pix cityscapes 256x256 GCC distill+darts iteration (optimize_parameters + '
                               'optimizer_netD_arch), student ngf32 / masked PatchGAN ndf128, teacher ngf64 / ndf128, '
                               'filter_cfgs=None (no pretrained checkpoint to prune)',
                   'batch_per_gpu': args.batch, 'global_batch': world * args.batch, 'image': '256x256',
                   'parallelism': 'dp%d' % world,
                   'dropout': 'on (counter RNG; the oracle / golden iterations run --no_dropout)'},
        'step_tflops': round(FLOP_PER_IMG * args.batch * args.steps / dt / 1e12, 2),
        'step_mfma_frac': round(FLOP_PER_IMG * args.batch * args.steps / dt / PEAK_BF16, 4),
        'loss_check': {k: round(v, 4) for k, v in losses.items()},
        'loss_check_step1': loss_check_step1(first_losses, args.batch, world),
        'launches_per_step': launches_per_step,
        'library': {'so': os.path.basename(_lib.LIB_PATH), 'tuning_hooks_at_default': options_default, 'tuning_hooks': option_vector,
                    'tile_plan_pinned_by_env': dict(ops._plan_pinned), 'tile_plan_of_the_step': ops.current_plan()},
    }
    if roof is not None:
        if 'conv_roofline' in roof:       # north_star: throughput as a fraction of the conv roofline
            roof['conv_roofline']['frac_of_step'] = round(roof['conv_roofline']['bound_ms'] / (1000.0 * dt / args.steps), 4)
        out['roofline'] = roof
    out['rccl_ranks'] = rccl_ranks
    if comm is not None:
        out['allreduce'] = comm
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
        n4 = cpu_baseline(iters=4, batch=4, budget=20.0)          # SURVEY 8(d) / BASELINE.md: N=1 and N=4
        out['cpu_baseline']['n4'] = {k: n4[k] for k in ('value', 'unit', 'sample')}
    if world == 1 and not args.no_other_configs:
        del model
        torch.cuda.empty_cache()
        out['other_configs'] = other_configs()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
