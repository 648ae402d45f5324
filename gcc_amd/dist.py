"""Data parallelism: one process per GPU, gradients summed over ranks on RCCL (torch.distributed
backend "nccl" on ROCm) at the backward boundaries of the iteration; BatchNorm statistics stay
rank-local (the reference has no SyncBN to mimic).  Initialised by the
launcher environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), e.g. torch.distributed.run."""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def _local_index():
    """LOCAL_RANK, folded onto the visible devices (one GPU per rank in production; a one-GPU rehearsal box maps every
    rank to cuda:0)"""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 1
    return int(os.environ.get('LOCAL_RANK', '0')) % max(n, 1)


def init_from_env(backend=None):
    """Initialise the process group when launched with WORLD_SIZE > 1 (idempotent)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 or is_dist():
        return world_size()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    # GCC_DIST_BACKEND=gloo: rehearsal rigs that put several ranks on one GPU (RCCL refuses duplicate devices)
    backend = backend or os.environ.get('GCC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(_local_index())
    dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
    return ws


def local_device(opt):
    """cuda:{LOCAL_RANK} under a multi-process launch, else cuda:{gpu_ids[0]} (models/Pix2Pix.py:356)"""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        dev = torch.device('cuda:%d' % _local_index())
    else:
        dev = torch.device('cuda:%d' % opt.gpu_ids[0])
    if torch.cuda.is_available():
        from . import ops
        torch.cuda.set_device(dev)           # one GPU per process: kernels go to this device's current stream
        ops.set_device_index(dev.index)
    return dev


def broadcast_module(module, src=0):
    """make replicas start identical (parameters and buffers)"""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def all_reduce_flat(flat_grads):
    """sum the flat gradient buffer of one parameter group over ranks (scaled by 1/world inside the
    Adam kernel via grad_scale)"""
    if is_dist():
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)


def all_reduce_grads(optimizer, async_op=False):
    """sum this optimizer's flat gradient bucket over ranks.  async_op: returns the work handle (the
    collective runs on RCCL's stream behind everything enqueued so far); call .wait() before the
    optimizer step."""
    if not is_dist():
        return None
    optimizer.set_grad_scale(1.0 / world_size())
    if async_op:
        return dist.all_reduce(optimizer.flat.grads, op=dist.ReduceOp.SUM, async_op=True)
    all_reduce_flat(optimizer.flat.grads)
    return None


class GradReducer:
    """Bucketed gradient all-reduce of one optimizer, overlapped with the backward pass that produces the gradients.

    The optimizer's flat gradient buffer is laid out in backward-completion order (engine.FlatParams(layout=...)): segment i
    = the parameters whose gradients are complete once the engine reports segment i.  Adjacent segments are coalesced into
    buckets of at least `bucket_bytes` (xGMI is point to point: a ring step moves bucket / world bytes per link, so buckets
    of a few tens of MB keep every link busy while staying well below the gradient of a whole network); when the last
    segment of a bucket is reported, its slice is all-reduced asynchronously (RCCL's own stream, ordered behind the stream
    the weight-gradient kernels of that segment were enqueued on) while the backward pass continues.  finish() makes the
    current stream wait for every bucket (and reduces whatever was never reported) -- call it before the Adam step.
    Sum over ranks; the 1/world factor is applied inside the Adam kernel."""

    def __init__(self, optimizer, bucket_bytes=32 << 20):
        self.opt = optimizer
        self.flat = optimizer.flat
        segs = self.flat.segments
        self.buckets, self.bucket_of = [], []
        b0 = 0
        for i, (b, e) in enumerate(segs):
            self.bucket_of.append(len(self.buckets))
            if (e - segs[b0][0]) * 4 >= bucket_bytes or i == len(segs) - 1:
                self.buckets.append((segs[b0][0], e, i))        # [begin, end) elements, last segment index
                b0 = i + 1
        self.handles = []
        self.launched = [False] * len(self.buckets)
        self.enabled = True

    def begin(self):
        """a new backward pass starts writing these gradients"""
        self.handles, self.launched = [], [False] * len(self.buckets)

    def segment_done(self, i, stream=None):
        """the kernels that complete segment i's gradients have been enqueued; `stream`: the stream they were enqueued on
        (default: the current stream)"""
        if not (self.enabled and is_dist()):
            return
        k = self.bucket_of[i]
        b, e, last = self.buckets[k]
        if i != last or self.launched[k]:
            return
        self.opt.set_grad_scale(1.0 / world_size())
        self.launched[k] = True
        if stream is not None:
            from . import ops
            with ops.on_stream(stream):
                self.handles.append(dist.all_reduce(self.flat.grads[b:e], op=dist.ReduceOp.SUM, async_op=True))
        else:
            self.handles.append(dist.all_reduce(self.flat.grads[b:e], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        if not is_dist():
            return
        self.opt.set_grad_scale(1.0 / world_size())
        for k, (b, e, _) in enumerate(self.buckets):
            if not self.launched[k]:
                self.launched[k] = True
                self.handles.append(dist.all_reduce(self.flat.grads[b:e], op=dist.ReduceOp.SUM, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles = []

    def wait(self):                 # the handle protocol of all_reduce_grads(async_op=True)
        self.finish()


class NativeComm:
    """The C ABI's own communicator (include/gcc_hip.h: gcc_comm_*), for hosts without a process group: RCCL directly, the id
    made by rank 0 (`NativeComm.unique_id()`) and handed to the other ranks by whatever channel the host has.  gcc_amd's
    model classes use torch.distributed (GradReducer above); this wrapper is the binding example of INTEGRATION.md and what
    the tests drive."""

    def __init__(self, rank, world, unique_id):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.gcc_comm_init(C.byref(self._h), int(rank), int(world), bytes(unique_id)), 'gcc_comm_init')
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().gcc_comm_unique_id(buf), 'gcc_comm_unique_id')
        return bytes(buf)

    def all_reduce_sum_(self, flat, stream=None):
        """in place, fp32, enqueued on `stream` (raw handle; default: the current stream)"""
        from . import _lib, ops
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.is_cuda
        _lib.check(self._lib.gcc_comm_allreduce_sum_f32(self._h, flat.data_ptr(), flat.numel(),
                                                        ops.stream() if stream is None else stream), 'gcc_comm_allreduce_sum_f32')
        return flat

    def close(self):
        if self._h:
            self._lib.gcc_comm_destroy(self._h)
            self._h = None


def all_reduce_sum(t):
    """sum over ranks of a small fp32 device vector, in place (the teacher's arch-difference terms: every replica must
    feed the same value into its EMA; the caller folds 1/world into its next kernel)"""
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def mean_dict(d, device):
    if not is_dist():
        return d
    keys = list(d.keys())
    t = torch.tensor([d[k] for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(t)
    t /= world_size()
    return type(d)((k, float(v)) for k, v in zip(keys, t.tolist()))


def shard_range(n_items, r=None, w=None):
    """contiguous [begin, end) share of n_items for rank r of w"""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per, rem = divmod(n_items, w)
    b = r * per + min(r, rem)
    return b, b + per + (1 if r < rem else 0)
