"""Data-parallel plumbing on CPU: world_size-2 gloo process groups (127.0.0.1)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gcc_amd import dist as gdist
    assert gdist.init_from_env(backend='gloo') == world
    assert gdist.world_size() == world and gdist.rank() == rank
    # gradient exchange: the sum over ranks of per-shard gradients, scaled by 1/world in the optimizer,
    # equals the full-batch mean gradient for any per-sample-mean loss (BatchNorm statistics excepted)
    torch.manual_seed(0)
    w = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    x = torch.randn(8, 3, dtype=torch.float64)
    y = torch.randn(8, 5, dtype=torch.float64)
    full = ((x @ w.t() - y) ** 2).mean()
    gfull = torch.autograd.grad(full, w)[0]
    b, e = gdist.shard_range(8)
    loc = ((x[b:e] @ w.t() - y[b:e]) ** 2).mean()
    g = torch.autograd.grad(loc, w)[0].clone()
    flat = g.reshape(-1).clone()
    gdist.all_reduce_flat(flat)
    ok_grad = torch.allclose(flat.reshape(5, 3) / world, gfull, atol=1e-12)
    # replicas start identical
    m = torch.nn.Linear(4, 4)
    torch.manual_seed(rank + 10)
    torch.nn.init.normal_(m.weight)
    gdist.broadcast_module(m)
    t = m.weight.detach().clone()
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    ok_bcast = all(torch.equal(gathered[0], gg) for gg in gathered)
    md = gdist.mean_dict({'a': float(rank), 'b': 2.0}, 'cpu')
    ok_mean = abs(md['a'] - (world - 1) / 2) < 1e-12 and md['b'] == 2.0
    # bucketed reducer over a flat gradient buffer in backward-completion order (U-Net module tree on CPU): segments reported
    # one by one launch their bucket's asynchronous all-reduce; after finish() the buffer holds the sum over ranks, exactly
    # what one all-reduce of the whole buffer gives, and unreported segments are swept up by finish()
    from gcc_amd import engine
    from gcc_amd.models.Pix2Pix import UnetGenertor
    torch.manual_seed(3)
    net = UnetGenertor(3, 3, 6, ngf=4)
    segs = engine.UnetEngine.grad_segments(net, 6)
    flatp = engine.FlatParams(list(net.parameters()), 'cpu', layout=segs)

    class _Opt:
        flat = flatp

        def set_grad_scale(self, s):
            self.scale = s
    opt_ = _Opt()
    red = gdist.GradReducer(opt_, bucket_bytes=4 << 10)
    gen = torch.Generator().manual_seed(100 + rank)
    flatp.grads.copy_(torch.randn(flatp.grads.shape, generator=gen))
    want = flatp.grads.clone()
    dist.all_reduce(want)
    red.begin()
    for i in range(len(segs) - 3):               # the last three segments are never reported
        red.segment_done(i)
    launched_early = sum(red.launched)
    red.finish()
    ok_buckets = (torch.equal(flatp.grads, want) and len(red.buckets) >= 3 and 0 < launched_early < len(red.buckets)
                  and abs(opt_.scale - 1.0 / world) < 1e-12 and all(red.launched))
    # a second pass reuses the reducer
    flatp.grads.copy_(torch.randn(flatp.grads.shape, generator=gen))
    want = flatp.grads.clone()
    dist.all_reduce(want)
    red.begin()
    for i in range(len(segs)):
        red.segment_done(i)
    red.finish()
    ok_buckets = ok_buckets and torch.equal(flatp.grads, want)
    t2 = torch.tensor([1.0 + rank, 2.0], dtype=torch.float32)
    gdist.all_reduce_sum(t2)
    ok_buckets = ok_buckets and t2.tolist() == [3.0, 4.0]
    # round 5: held buckets -- a reducer that is disabled during the backward pass issues nothing (the online teacher's generator:
    # its buckets go out later, behind the student's discriminator buckets) and finish() reduces everything; the Deferred handle of
    # the un-bucketed form; a communicator of its own for a concurrent chain (off unless GCC_DP_CHAIN_GROUPS=1)
    flatp.grads.copy_(torch.randn(flatp.grads.shape, generator=gen))
    want = flatp.grads.clone()
    dist.all_reduce(want)
    red.begin()
    red.enabled = False
    for i in range(len(segs)):
        red.segment_done(i)
    held = sum(red.launched) == 0
    red.enabled = True
    red.finish()
    ok_buckets = ok_buckets and held and torch.equal(flatp.grads, want) and all(red.launched)
    flatp.grads.copy_(torch.randn(flatp.grads.shape, generator=gen))
    want = flatp.grads.clone()
    dist.all_reduce(want)
    h = gdist.Deferred(opt_)
    still = not torch.equal(flatp.grads, want)
    h.wait()
    ok_buckets = ok_buckets and still and torch.equal(flatp.grads, want)
    assert gdist.chain_group('teacher') is None
    os.environ['GCC_DP_CHAIN_GROUPS'] = '1'
    os.environ['GCC_DP_BF16'] = '1'
    # experimental variants are ignored (with a warning) unless the ONE experimental switch is on as well
    assert gdist.chain_group('teacher') is None and not gdist.bf16_buckets() and not gdist.experimental()
    os.environ['GCC_DP_EXPERIMENTAL'] = '1'
    assert gdist.bf16_buckets()
    os.environ['GCC_DP_BF16'] = '0'
    grp = gdist.chain_group('teacher')
    ok_buckets = ok_buckets and grp is not None and gdist.chain_group('teacher') is grp
    red.set_group(grp)
    flatp.grads.copy_(torch.randn(flatp.grads.shape, generator=gen))
    want = flatp.grads.clone()
    dist.all_reduce(want)
    red.begin()
    red.finish()
    ok_buckets = ok_buckets and torch.equal(flatp.grads, want)
    ok_buckets = ok_buckets and gdist.rccl_ranks() == world
    q.put((rank, ok_grad, ok_bcast, ok_mean and ok_buckets, (b, e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gloo_world2_gradient_exchange():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in range(2)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    res.sort()
    assert [r[4] for r in res] == [(0, 4), (4, 8)]
    for r in res:
        assert r[1] and r[2] and r[3], r


def test_shard_range_partitions():
    from gcc_amd import dist as gdist
    for n in (1, 7, 8, 16, 17):
        for w in (1, 2, 3, 8):
            parts = [gdist.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for a, b in zip(parts[:-1], parts[1:]):
                assert a[1] == b[0]
