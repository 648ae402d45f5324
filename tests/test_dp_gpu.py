"""Data-parallel path end to end on the GPU box: two ranks (gloo process group, both on cuda:0 because
the test box has one GPU; the production backend is nccl = RCCL, one GPU per rank) run the real
Pix2Pix GCC iteration on different shards.  Checks: replicas start identical, gradients are exchanged
(replicas stay bit-identical after the step although their inputs differ), and the result equals a
single-process run that is fed the rank-averaged gradients (grad_scale = 1/world in the Adam kernel)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(q, procs, n, timeout=1200):
    """n results from the workers' queue; fails as soon as a worker has died instead of waiting out the timeout"""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, 'a worker process died (exit codes %s): see its traceback above' % dead
            assert time.time() - t0 < timeout, 'workers timed out'
    return out


def _worker(rank, world, port, q, backend='gloo', iters=1, buckets='1'):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank) if backend == 'nccl' else '0', HSA_ENABLE_IPC_MODE_LEGACY='0',
                      GCC_DP_BUCKETS=buckets)
    import torch.distributed as dist
    from gcc_amd import dist as gdist
    gdist.init_from_env(backend=backend)
    from tests.test_pix2pix_gpu import GCC_ARGV, build_model
    torch.manual_seed(100 + rank)            # different initial weights per rank: the broadcast must fix that
    model, teacher, opt = build_model(GCC_ARGV, teacher_ndf=16)
    model.model_train()
    sd0 = model.netG.state_dict()['model.model.0.weight'].float().cpu().clone()
    g = torch.Generator().manual_seed(7 + rank)     # different data per rank
    A, B = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    for _ in range(iters):
        model.set_input({'A': A, 'B': B, 'A_paths': [''], 'B_paths': ['']})
        model.optimize_parameters()
        model.set_input({'A': B, 'B': A, 'A_paths': [''], 'B_paths': ['']})
        model.clipping_mask_alpha()
        model.optimizer_netD_arch()
    model.finish_G_update()
    teacher.finish_G_update()
    torch.cuda.synchronize()
    out = {'rank': rank, 'w0': sd0.numpy(), 'bucketed': model.optimizer_D.reducer is not None}
    for name, mod in (('sG', model.netG), ('sD', model.netD), ('tG', teacher.netG), ('tD', teacher.netD)):
        out[name] = torch.cat([v.detach().float().cpu().reshape(-1) for k, v in mod.state_dict().items()
                               if k.endswith('weight') or k.endswith('bias') or k.endswith('alpha')]).numpy()
    # everything optimizer_G owns, the 1x1 distillation transform convs included (they live outside netG / netD)
    out['flatG'] = model.optimizer_G.flat.values.detach().float().cpu().numpy()
    out['T'] = torch.cat([t.weight.detach().float().cpu().reshape(-1) for t in model.transform_convs]).numpy()
    out['losses'] = dict(model.get_current_losses())
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(buckets):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 'gloo', 1, buckets)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs, 2), key=lambda d: d['rank'])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(1500)       # the spawned ranks import torch afresh: minutes on a cold box, seconds otherwise
def test_two_ranks_stay_identical():
    import numpy as np
    a, b = _two_ranks('1')
    assert a['bucketed'] and b['bucketed'], 'the bucketed reducer (and its self-check) should be on'
    assert np.array_equal(a['w0'], b['w0']), 'replicas did not start from the same weights'
    for k in ('sG', 'sD', 'tG', 'tD', 'T', 'flatG'):
        assert np.array_equal(a[k], b[k]), 'replicas diverged in %s: gradients were not exchanged identically' % k
        assert np.isfinite(a[k]).all()
    assert a['losses'] == b['losses']          # logged losses are rank-averaged
    # the bucketed, overlapped exchange against the plain flat all-reduce per optimizer on the same shards -- a
    # missed ordering between the weight-gradient kernels and a bucket's all-reduce would leave the replicas identical to
    # each other and different from this
    c, d = _two_ranks('0')
    assert not c['bucketed']
    for k in ('sG', 'sD', 'tG', 'tD', 'T', 'flatG'):
        assert np.array_equal(a[k], c[k]), 'bucketed and flat gradient exchange disagree in %s' % k
    assert a['losses'] == c['losses']


@pytest.mark.timeout(1500)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: one RCCL rank per device')
def test_two_rccl_ranks_two_devices():
    """the production backend on real devices (skipped on the one-GPU test box; the first box with two GPUs runs it): two
    RCCL ranks, one GPU each, two whole iterations on different shards -- bucketed all-reduces launched during the
    backward, the asynchronous teacher-generator exchange, the summed teacher arch difference -- replicas bit-identical"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 'nccl', 2)) for r in range(2)]
    for p in procs:
        p.start()
    a, b = sorted(_collect(q, procs, 2), key=lambda d: d['rank'])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    import numpy as np
    for k in ('sG', 'sD', 'tG', 'tD', 'T', 'flatG'):
        assert np.array_equal(a[k], b[k]), 'replicas diverged in %s' % k
        assert np.isfinite(a[k]).all()
    assert a['losses'] == b['losses']


def _native_worker(rank, world, uid, q):
    sys.path.insert(0, ROOT)
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    import torch as th
    from gcc_amd import dist as gdist
    from gcc_amd import ops
    th.cuda.set_device(rank)
    ops.set_device_index(rank)
    comm = gdist.NativeComm(rank, world, uid)
    assert comm.count() == world, (comm.count(), world)          # ncclCommCount, not the launcher's word
    g = th.Generator().manual_seed(3 + rank)
    host = th.randn(1 << 20, generator=g)
    buf = host.to('cuda:%d' % rank)
    side = th.cuda.Stream()
    with th.cuda.stream(side):                  # ordered on the caller's stream like a kernel
        buf.mul_(2.0)
        comm.all_reduce_sum_(buf)
        buf.mul_(0.5)
    th.cuda.synchronize()
    q.put({'rank': rank, 'sum': buf.cpu().numpy(), 'own': host.numpy()})
    comm.close()


@pytest.mark.timeout(900)
def test_native_comm_single_rank():
    """include/gcc_hip.h gcc_comm_*: communicator of one rank on this box's GPU -- init, an all-reduce enqueued between two
    kernels of a side stream, destroy; the sum over one rank is the buffer itself"""
    from gcc_amd import dist as gdist
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_native_worker, args=(0, 1, gdist.NativeComm.unique_id(), q))
    p.start()
    res = _collect(q, [p], 1)[0]
    p.join(120)
    assert p.exitcode == 0
    import numpy as np
    assert np.array_equal(res['sum'], res['own'] * np.float32(2.0) * np.float32(0.5))


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs: one RCCL rank per device')
def test_native_comm_two_ranks_two_devices():
    from gcc_amd import dist as gdist
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    uid = gdist.NativeComm.unique_id()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, uid, q)) for r in range(2)]
    for p in procs:
        p.start()
    a, b = sorted(_collect(q, procs, 2), key=lambda d: d['rank'])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    import numpy as np
    want = (a['own'] * np.float32(2.0) + b['own'] * np.float32(2.0)) * np.float32(0.5)
    assert np.array_equal(a['sum'], b['sum']) and np.allclose(a['sum'], want, rtol=0, atol=1e-6)


def _rccl_worker(port, q, route='torch'):
    sys.path.insert(0, ROOT)
    # GCC_DP_FORCE_BUCKETS: the bucketed reducer (and its self-check) also with the one rank this box can give RCCL
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', GCC_DP_FORCE_BUCKETS='1', GCC_DP_COMM=route)
    import torch.distributed as dist
    from tests.test_pix2pix_gpu import GCC_ARGV, build_model

    def run(use_dist):
        torch.manual_seed(5)
        model, teacher, opt = build_model(GCC_ARGV, teacher_ndf=16)
        if use_dist:
            model._world = teacher._world = 2        # take the data-parallel code paths (async teacher-G bucket included)
        model.model_train()
        g = torch.Generator().manual_seed(11)
        out = []
        for it in range(2):
            A, B = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
            model.set_input({'A': A, 'B': B, 'A_paths': [''], 'B_paths': ['']})
            model.optimize_parameters()
            model.set_input({'A': B, 'B': A, 'A_paths': [''], 'B_paths': ['']})
            model.clipping_mask_alpha()
            model.optimizer_netD_arch()
            torch.cuda.synchronize()
            out.append(dict(model.get_current_losses()))
        w = torch.cat([p.detach().float().reshape(-1) for p in list(model.netG.parameters()) + list(teacher.netG.parameters())]).cpu()
        return out, w, (model.optimizer_D.reducer is not None and model.optimizer_D.reducer.route)
    ref_losses, ref_w, _ = run(False)
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    losses, w, bucket_route = run(True)
    dist.barrier()
    dist.destroy_process_group()
    q.put({'ref': ref_losses, 'got': losses, 'same_weights': bool(torch.equal(ref_w, w)), 'bucket_route': bucket_route})


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('route', ['torch', 'native'])
def test_rccl_single_rank_with_teacher_stream(route):
    """the nccl (= RCCL) backend itself, one rank: the BUCKETED all-reduces issued from the weight-gradient side streams of the
    main and the teacher stream (after the self-check of that ordering on this backend), the asynchronous teacher-generator
    bucket and its late wait -- through torch.distributed and through the C ABI's own communicator (GCC_DP_COMM=native:
    gcc_comm_allreduce_sum_f32 on the side stream itself) -- results must equal the run without a process group"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q, route))
    p.start()
    res = _collect(q, [p], 1)[0]
    p.join(120)
    assert p.exitcode == 0
    assert res['bucket_route'] == route, res['bucket_route']
    assert res['got'] == res['ref'], (res['got'], res['ref'])
    assert res['same_weights']


def _rccl_replay_worker(port, q, bf16):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0', GCC_DP_FORCE_BUCKETS='1', GCC_DP_COMM='native', GCC_DP_BF16='1' if bf16 else '0',
                      GCC_DP_EXPERIMENTAL='1')          # replay with collectives and bf16 buckets: experimental variants (dist.experimental)
    import torch.distributed as dist
    from gcc_amd.replay import IterationReplay
    from tests.test_pix2pix_gpu import GCC_ARGV, build_model
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)

    def run(enabled):
        torch.manual_seed(5)
        model, teacher, opt = build_model(GCC_ARGV, teacher_ndf=16)        # --no_dropout: replayable
        model._world = teacher._world = 2            # take the data-parallel code paths (async teacher-G bucket included)
        model.model_train()
        g = torch.Generator().manual_seed(11)
        data = [{'A': torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, 'B': torch.rand(2, 3, 64, 64, generator=g) * 2 - 1,
                 'A_paths': [''], 'B_paths': ['']} for _ in range(3)]
        rp = IterationReplay(model, opt, warmup=2, threads=4, enabled=enabled)
        modes, losses = [], []
        for i in range(6):
            modes.append(rp.step(data[i % 3], data[(i + 1) % 3]))
            losses.append(dict(model.get_current_losses()))
        model.finish_G_update()
        teacher.finish_G_update()
        torch.cuda.synchronize()
        info = rp.info()
        w = torch.cat([p.detach().float().reshape(-1) for m in (model.netG, model.netD, teacher.netG, teacher.netD)
                       for p in m.parameters()]).cpu()
        rp.invalidate()
        return modes, losses, w, info, (model.optimizer_D.reducer is not None and model.optimizer_D.reducer.route)
    m0, l0, w0, _, route = run(False)
    m1, l1, w1, info, _ = run(True)
    dist.barrier()
    dist.destroy_process_group()
    q.put({'eager_modes': m0, 'modes': m1, 'same_losses': l0 == l1, 'same_weights': bool(torch.equal(w0, w1)), 'info': info,
           'route': route, 'finite': bool(torch.isfinite(w1).all())})


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('bf16', [False, True])
def test_replay_composes_with_data_parallel_native_route(bf16):
    """with the gradient exchange on the C ABI's own communicator (the default route on an RCCL process group)
    the all-reduces are part of the launch recording -- bucketed reducer on the weight-gradient side streams, the
    asynchronous teacher-generator bucket, the summed arch-difference terms -- and a replayed data-parallel iteration leaves
    the bits of the eager one (one RCCL rank: what this box can give; issued from ONE host thread, whatever was asked for).
    bf16: the same with the buckets cast to bf16 for the exchange (GCC_DP_BF16=1)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_replay_worker, args=(_free_port(), q, bf16))
    p.start()
    res = _collect(q, [p], 1)[0]
    p.join(120)
    assert p.exitcode == 0
    assert res['route'] == 'native', res['route']
    assert set(res['eager_modes']) == {'eager'} and res['modes'] == ['eager', 'eager', 'record', 'replay', 'replay', 'replay'], res
    assert res['info']['threads'] == 1 and res['info']['launches'] > 100, res['info']
    assert res['same_losses'] and res['same_weights'] and res['finite']


@pytest.mark.timeout(1500)
def test_bench_two_ranks_driver_command_line():
    """the driver's multi-GPU invocation of bench.py, rehearsed with two ranks on this box's one GPU (gloo in place of
    RCCL, which refuses duplicate devices): barrier / max-over-ranks timing / rank-0 JSON line"""
    import json
    import subprocess
    env = dict(os.environ, GCC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['config']['global_batch'] == 32 and d['scaling'] == 'weak'
    assert d['value'] > 0 and abs(d['value'] - 32 * 2 / (d['ms_per_step'] * 2 / 1000.0)) < 0.5
    assert 'roofline' in d
