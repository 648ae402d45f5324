"""Launch replay (gcc_amd.replay, gcc_replay_* of the C-ABI): an iteration recorded once and re-issued from native code -- on one
thread and with one host thread per HIP stream -- leaves every parameter, optimizer moment and logged loss BIT-identical to the
eager host path, over iterations whose Adam factors and image-pool draws differ from the recorded one."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _cyclegan():
    from tests.conftest import GOLDEN
    from tests.test_pix2pix_gpu import load
    from tests.test_cyclegan_gpu import _build
    z = load(GOLDEN, 'cyclegan_gcc.npz')
    model, teacher, opt = _build(z)
    g = torch.Generator().manual_seed(5)
    data = [{'A': torch.rand(1, 3, 64, 64, generator=g) * 2 - 1, 'B': torch.rand(1, 3, 64, 64, generator=g) * 2 - 1,
             'A_paths': ['a'], 'B_paths': ['b']} for _ in range(4)]
    return model, teacher, opt, data


def _sagan():
    from tests.conftest import GOLDEN
    from tests.test_pix2pix_gpu import load
    from tests.test_sagan_gpu import _build
    z = load(GOLDEN, 'sagan_gcc.npz')
    model, teacher, opt = _build(z)
    g = torch.Generator().manual_seed(6)
    b = opt.batch_size
    data = [{'z': torch.randn(b, opt.z_dim, generator=g), 'real_img': torch.rand(b, 3, 64, 64, generator=g) * 2 - 1, 'img_path': [''] * b}
            for _ in range(4)]
    return model, teacher, opt, data


def _srgan():
    from tests.conftest import GOLDEN
    from tests.test_pix2pix_gpu import load
    from tests.test_srgan_gpu import _build
    z = load(GOLDEN, 'srgan_gcc.npz')
    model, teacher, opt = _build(z)
    g = torch.Generator().manual_seed(7)
    data = [{'lr': torch.rand(2, 3, 24, 24, generator=g) * 2 - 1, 'hr': torch.rand(2, 3, 96, 96, generator=g) * 2 - 1,
             'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2} for _ in range(4)]
    return model, teacher, opt, data


def _pix2pix():
    """Pix2Pix GCC with --no_dropout (dropout seeds are by-value arguments nothing patches: Pix2PixModel.replay_supported)"""
    from tests.conftest import GOLDEN
    from tests.test_pix2pix_gpu import load, _build_gcc
    z = load(GOLDEN, 'pix2pix_gcc_d6.npz')
    model, teacher, opt = _build_gcc(z)
    assert model.replay_supported
    g = torch.Generator().manual_seed(8)
    data = [{'A': torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, 'B': torch.rand(2, 3, 64, 64, generator=g) * 2 - 1,
             'A_paths': ['a'] * 2, 'B_paths': ['b'] * 2} for _ in range(4)]
    return model, teacher, opt, data


def _state(model, teacher):
    out = {}
    for tag, m in (('s', model), ('t', teacher)):
        for k, v in m.state_dict().items():
            out['%s.%s' % (tag, k)] = v.detach().clone()
        for name in dir(m):
            o = getattr(m, name, None)
            for pn in ('plan', 'plan_dup'):
                p = getattr(o, pn, None) if name.startswith('optimizer') else None
                if p is not None:
                    for i, (a, b) in enumerate(zip(p.m, p.v)):
                        out['%s.%s.%s.m%d' % (tag, name, pn, i)] = a.detach().clone()
                        out['%s.%s.%s.v%d' % (tag, name, pn, i)] = b.detach().clone()
    return out


def _run(build, threads, enabled, iters=7):
    from gcc_amd.replay import IterationReplay
    random.seed(99)
    torch.manual_seed(3)
    model, teacher, opt, data = build()
    rp = IterationReplay(model, opt, warmup=2, threads=threads, enabled=enabled)
    modes = []
    losses = []
    for i in range(iters):
        modes.append(rp.step(data[i % 4], data[(i + 1) % 4]))
        losses.append(dict(model.get_current_losses()))
    torch.cuda.synchronize()
    info = rp.info()
    st = _state(model, teacher)
    rp.invalidate()
    return modes, losses, st, info


@pytest.mark.parametrize('which', ['cyclegan', 'sagan', 'srgan', 'pix2pix'])
def test_replay_is_bit_identical_to_the_eager_iteration(which):
    build = {'cyclegan': _cyclegan, 'sagan': _sagan, 'srgan': _srgan, 'pix2pix': _pix2pix}[which]
    m0, l0, s0, _ = _run(build, 1, False)
    assert set(m0) == {'eager'}
    for threads in (1, 4):
        m1, l1, s1, info = _run(build, threads, True)
        assert m1 == ['eager', 'eager', 'record'] + ['replay'] * 4, m1
        assert info['launches'] > 100 and info['streams'] >= 2 and info['threads'] == min(threads, info['streams']), info
        print(which, 'threads', threads, info)
        assert l1 == l0, 'logged losses differ'
        assert s0.keys() == s1.keys() and len(s0) > 20
        bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not bad, 'differ after replay: %s' % bad[:8]


def test_replay_invalidate_and_shape_change():
    """a changed batch shape or invalidate() drops the recording; the next steps run eagerly once and record again"""
    from gcc_amd.replay import IterationReplay
    random.seed(1)
    model, teacher, opt, data = _srgan()
    rp = IterationReplay(model, opt, warmup=1, threads=1, enabled=True)
    modes = [rp.step(data[i % 4], data[(i + 1) % 4]) for i in range(4)]
    assert modes == ['eager', 'record', 'replay', 'replay']
    rp.invalidate()
    modes = [rp.step(data[i % 4], data[(i + 1) % 4]) for i in range(3)]
    assert modes == ['record', 'replay', 'replay'] or modes == ['eager', 'record', 'replay'], modes
    torch.cuda.synchronize()


def test_teacher_enqueue_thread_changes_nothing(monkeypatch):
    """GCC_TEACHER_THREAD=1 (the teacher's step enqueued by a second host thread, models/_streams.py): same bits"""
    m0, l0, s0, _ = _run(_cyclegan, 1, False, iters=3)
    monkeypatch.setenv('GCC_TEACHER_THREAD', '1')
    m1, l1, s1, _ = _run(_cyclegan, 1, False, iters=3)
    assert l0 == l1
    bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    assert not bad, bad[:8]


def test_pix2pix_stream_forks_change_nothing(monkeypatch):
    """the two stretches of the Pix2Pix step that run independent chains side by side on the auxiliary stream (round 4:
    Pix2Pix.DISTILL_FORK -- the distillation terms on the generator's features beside the teacher discriminator's pass over
    the student's fake; ARCH_FORK -- the architecture step's two discriminator backward passes, the second in gradient buffers
    of its own, its alpha gradients added afterwards) against the in-line order: every weight, optimizer moment, BatchNorm
    statistic and logged loss bit for bit after five iterations"""
    from gcc_amd.models import Pix2Pix as P
    monkeypatch.setattr(P, 'DISTILL_FORK', False)
    monkeypatch.setattr(P, 'ARCH_FORK', False)
    monkeypatch.setattr(P, 'ARCH_EARLY', False)
    m0, l0, s0, _ = _run(_pix2pix, 1, False, iters=5)
    # (ARCH_EARLY: the online teacher's architecture-step forward started before the student has finished reading the teacher:
    # second set of generator activations, deferred BatchNorm running updates of the teacher's discriminator)
    monkeypatch.setattr(P, 'ARCH_FREE_EARLY', False)
    for fork in ((True, False, False, False), (False, True, False, False), (False, False, True, False), (False, False, False, True),
                 (True, True, True, True)):
        # (fork[3], ARCH_FREE_EARLY: the teacher's stream copies its difference scalar itself and is released behind its own
        # architecture-step part)
        monkeypatch.setattr(P, 'DISTILL_FORK', fork[0])
        monkeypatch.setattr(P, 'ARCH_FORK', fork[1])
        monkeypatch.setattr(P, 'ARCH_EARLY', fork[2])
        monkeypatch.setattr(P, 'ARCH_FREE_EARLY', fork[3])
        m1, l1, s1, _ = _run(_pix2pix, 1, False, iters=5)
        assert l0 == l1, 'logged losses differ with forks %s' % (fork,)
        bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not bad, 'forks %s: %s' % (fork, bad[:8])


def test_sagan_distill_fork_changes_nothing(monkeypatch):
    """SAGAN.G_FORK: backward_G's distillation block on the auxiliary stream beside the discriminator's pass -- eager and
    replayed -- against the in-line order: same bits"""
    from gcc_amd.models import SAGAN as Sa
    monkeypatch.setattr(Sa, 'G_FORK', False)
    m0, l0, s0, _ = _run(_sagan, 1, False, iters=5)
    monkeypatch.setattr(Sa, 'G_FORK', True)
    for enabled, threads in ((False, 1), (True, 4)):
        m1, l1, s1, info = _run(_sagan, threads, enabled, iters=5)
        assert l0 == l1, 'logged losses differ (replay %s)' % enabled
        bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not bad, 'fork (replay %s): %s' % (enabled, bad[:8])


def test_srgan_vgg_fork_changes_nothing(monkeypatch):
    """SRGAN.SR_FORK: backward_G's VGG chain on the auxiliary stream beside the discriminator's pass -- eager and replayed --
    against the in-line order: same bits"""
    from gcc_amd.models import SRGAN as Sr
    monkeypatch.setattr(Sr, 'SR_FORK', False)
    m0, l0, s0, _ = _run(_srgan, 1, False, iters=5)
    monkeypatch.setattr(Sr, 'SR_FORK', True)
    for enabled, threads in ((False, 1), (True, 4)):
        m1, l1, s1, info = _run(_srgan, threads, enabled, iters=5)
        assert l0 == l1, 'logged losses differ (replay %s)' % enabled
        bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not bad, 'fork (replay %s): %s' % (enabled, bad[:8])


def test_cyclegan_two_sides_fork_changes_nothing(monkeypatch):
    """CycleGAN.CYCLE_FORK: side B of forward / backward_G / backward_D / the architecture step on the auxiliary stream beside
    side A -- eager and replayed -- against the in-line order: every weight, optimizer moment and logged loss bit for bit"""
    from gcc_amd.models import CycleGAN as Cg
    monkeypatch.setattr(Cg, 'CYCLE_FORK', 0)
    m0, l0, s0, _ = _run(_cyclegan, 1, False, iters=5)
    for mode, enabled, threads in ((1, False, 1), (1, True, 4), (2, False, 1), (2, True, 4)):
        # 1: the student's sides, weight gradients on side streams; 2: the teacher's sides too, weight gradients on their chains
        monkeypatch.setattr(Cg, 'CYCLE_FORK', mode)
        m1, l1, s1, info = _run(_cyclegan, threads, enabled, iters=5)
        if enabled:
            assert m1[-1] == 'replay' and info['streams'] >= 3, (m1, info)
        assert l0 == l1, 'logged losses differ (mode %d, replay %s)' % (mode, enabled)
        bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
        assert not bad, 'fork mode %d (replay %s): %s' % (mode, enabled, bad[:8])


@pytest.mark.parametrize('which', ['srgan', 'cyclegan'])
def test_train_loop_with_replay_ends_on_the_same_weights(tmp_path, monkeypatch, which):
    """python -m gcc_amd.train with GCC_REPLAY=1 (the iteration recorded, replayed, dropped at the epoch boundary with the new
    learning rate and recorded again) against the eager loop: same seeds, same synthetic batches, three epochs -- every saved
    tensor of the final checkpoint is bit-identical"""
    import os
    from gcc_amd import train
    os.environ['GCC_VGG19_RANDOM'] = '1'
    common = ['--dataroot', 'synthetic:4', '--gpu_ids', '0', '--online_distillation', '--darts_discriminator', '--n_epochs', '2',
              '--n_epochs_decay', '1', '--print_freq', '100', '--ngf', '8', '--ndf', '8', '--teacher_ngf', '16']
    if which == 'srgan':
        argv = common + ['--model', 'srgan', '--image_size', '48', '--batch_size', '2']
    else:
        argv = common + ['--model', 'cyclegan', '--crop_size', '64', '--batch_size', '1']
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('GCC_REPLAY', mode)
        random.seed(7)
        torch.manual_seed(7)
        model = train.main(argv + ['--checkpoints_dir', str(tmp_path / ('ck' + mode)), '--name', 'r'])
        torch.cuda.synchronize()
        ck = tmp_path / ('ck' + mode) / 'r' / 'checkpoints'
        files = sorted(os.listdir(str(ck)))
        assert files, 'no checkpoint written'
        out[mode] = torch.load(str(ck / files[-1]), map_location='cpu')
        del model

    def flat(prefix, o, acc):
        if torch.is_tensor(o):
            acc[prefix] = o
        elif isinstance(o, dict):
            for k, v in o.items():
                flat('%s.%s' % (prefix, k), v, acc)
        elif isinstance(o, (list, tuple)):
            for i, v in enumerate(o):
                flat('%s[%d]' % (prefix, i), v, acc)
        return acc
    a, b = flat('', out['0'], {}), flat('', out['1'], {})
    assert a.keys() == b.keys() and len(a) > 20
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, bad[:8]


def test_replay_stages_loader_batches_behind_their_ready_event():
    """batches that come from gcc_amd.data's GPU loaders are written on the LOADER's stream and carry a 'ready' event.
    IterationReplay._stage must wait for it before copying them into its persistent buffers (and must not forward it: the
    staged buffers are produced by the copies, set_input records its own event behind them).  The producer here is made slow
    on purpose (large fills in front of the batch on its stream): a staging copy that did not wait would read the old bytes."""
    from gcc_amd.data import _GpuFileLoader
    from gcc_amd.replay import IterationReplay
    random.seed(5)
    torch.manual_seed(5)
    model, teacher, opt, data = _srgan()
    loader = _GpuFileLoader.__new__(_GpuFileLoader)
    loader._stream = None
    junk = torch.empty(128 << 20, dtype=torch.float32, device=DEV)

    def produce(d):
        def fn():
            for i in range(6):
                junk.fill_(float(i))                          # ~ms of work in front of the batch on the loader's stream
            return {k: (v.to(DEV, non_blocking=True) if torch.is_tensor(v) else v) for k, v in d.items()}
        return loader._produce(fn)
    rp = IterationReplay(model, opt, warmup=1, threads=1, enabled=True)
    modes, losses = [], []
    for i in range(5):
        b, vb = produce(data[i % 4]), produce(data[(i + 1) % 4])
        assert 'ready' in b
        modes.append(rp.step(b, vb))
        assert 'ready' not in rp.static[0] and 'ready' not in rp.static[1]
        for k in ('lr', 'hr'):                                # the staged copy holds the batch, not what was there before
            assert torch.equal(rp.static[0][k].cpu(), data[i % 4][k])
        losses.append(dict(model.get_current_losses()))
    assert modes == ['eager', 'record', 'replay', 'replay', 'replay'], modes
    # the same five iterations from host batches, eagerly: same losses
    random.seed(5)
    torch.manual_seed(5)
    model2, teacher2, opt2, data2 = _srgan()
    rp2 = IterationReplay(model2, opt2, warmup=1, threads=1, enabled=False)
    for i in range(5):
        rp2.step(data2[i % 4], data2[(i + 1) % 4])
        assert dict(model2.get_current_losses()) == losses[i], i
    rp.invalidate()
