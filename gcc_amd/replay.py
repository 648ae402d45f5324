"""Launch replay of a training iteration (include/gcc_hip.h: gcc_replay_*; gcc_amd/csrc/replay.hip).

The reference's loop (train.py:128-140) calls model.set_input / optimize_parameters / optimizer_netD_arch per batch and leaves
the rest to PyTorch's eager dispatcher.  Here an iteration is a fixed sequence of a few thousand launches of libgcc_hip.so,
and for the small models (CycleGAN at batch 1, SAGAN 64 x 64, SRGAN 96 x 96 crops) the HOST's launch rate bounds the iteration:
~2.5 us of Python and ~4.3 us of hipLaunchKernel per launch on one thread, against kernels of 5-20 us that two or three streams
run side by side.  IterationReplay runs a few iterations eagerly, records one (every launch, memset / copy, event record /
wait of the calling thread, with its argument values) and from then on re-issues the recording from native code, each HIP
stream's share from its own host thread.

What makes the iteration replayable:
  * inputs enter through persistent device buffers (the batch is copied into them before every iteration);
  * the recorded iteration allocates from a private torch.cuda.MemPool that stays alive with the recording, so every address
    baked into it keeps its meaning (activations, workspaces and optimizer state were allocated before and persist anyway);
  * by-value arguments that change per iteration are patched before every run (ops.note_dynamic: Adam's bias corrections,
    the CycleGAN image pool's random draws);
  * anything else that changes a launch argument invalidates the recording: call invalidate() after update_learning_rate(),
    adaptive_ema_beta(), pruning, a changed batch shape (step() checks shapes itself); the next step() records again.
Data parallelism: the gradient all-reduces of the C ABI's own communicator (dist.comm_route() 'native': GCC_DP_COMM=native, opt-in
since round 5) are recorded like launches and replayed from ONE host thread in the recorded order (the same on every rank);
with the exchange on torch.distributed step() stays eager.  So does a model class that says `replay_supported = False`:
Pix2Pix with dropout on (the dropout seeds are by-value launch arguments that nothing patches -- and its iteration is bound by
its convolutions, not by the host: the eager host enqueues a step in 5.8 ms and runs ahead of the 15.6 ms the device needs
(`profiles/r4v_host_enqueue_unblocked.txt`); replayed -- GCC_REPLAY_FORCE=1, timing only -- it measured 17.9 ms,
`profiles/r4u_replay_pix2pix.txt`).
"""
import ctypes as C
import os

import torch

from . import ops


class IterationReplay:
    """step(batch, val_batch) == the loop body of gcc_amd.train.main:
           model.set_input(batch); model.optimize_parameters()
           model.set_input(val_batch); model.clipping_mask_alpha(); model.optimizer_netD_arch()      (darts + online teacher)
    """

    def __init__(self, model, opt, warmup=3, threads=None, enabled=None):
        self.model, self.opt = model, opt
        self.warmup = warmup
        self.threads = int(os.environ.get('GCC_REPLAY_THREADS', '4')) if threads is None else threads
        self.enabled = (os.environ.get('GCC_REPLAY', '1') != '0') if enabled is None else enabled
        self.arch = bool(getattr(opt, 'darts_discriminator', False)) and getattr(model, 'teacher_model', None) is not None
        self.seen = 0
        self.rec = None             # gcc_replay_t*
        self.pool = None            # torch.cuda.MemPool of the recorded iteration
        self.dynamic = []
        self.static = [{}, {}]      # persistent device copies of the batch / validation batch tensors
        self.shapes = None
        self.replayed = 0

    # ------------------------------------------------------------------------------------------------
    def _eager(self, a, b):
        m = self.model
        m.set_input(a)
        m.optimize_parameters()
        if self.arch:
            m.set_input(b)
            m.clipping_mask_alpha()
            m.optimizer_netD_arch()

    def _stage(self, which, batch):
        """the batch's tensors copied into persistent device buffers (allocated on first sight of a key / shape); returns the
        batch dict the model sees: the same object every iteration.

        A device-resident batch may carry 'ready' (a torch.cuda.Event its producer -- gcc_amd.data's loader stream -- recorded
        behind the last kernel that wrote it): the copies below wait for it on the current stream, and the source tensors are
        marked as used on that stream so that the caching allocator does not hand their blocks back to the producer while a
        copy is pending (without either the copy could read a half-written batch).  'ready' is NOT forwarded: the
        staged buffers are produced by the copies on the current stream, so set_input records its own event behind them
        (models/_streams._note_input) and the teacher's stream waits for THAT -- an ops.Event, hence part of the recording."""
        if batch is None:
            return None
        st = self.static[which]
        dev = self.model.device
        cur = ops.current_stream()
        ready = batch.get('ready') if hasattr(batch, 'get') else None
        if ready is not None:
            ops.wait_event(cur, ready)
        for k, v in batch.items():
            if k == 'ready':
                continue
            if torch.is_tensor(v):
                t = st.get(k)
                if t is None or t.shape != v.shape or t.dtype != v.dtype:
                    t = st[k] = torch.empty(v.shape, dtype=v.dtype, device=dev)
                t.copy_(v, non_blocking=True)
                if v.is_cuda:
                    v.record_stream(cur)
            else:
                st[k] = v
        st.pop('ready', None)
        return st

    def _signature(self, a, b):
        return tuple((k, tuple(v.shape)) for d in (a, b) if d is not None for k, v in sorted(d.items(), key=lambda kv: kv[0]) if torch.is_tensor(v))

    def usable(self):
        if not self.enabled or not torch.cuda.is_available():
            return False
        if not getattr(self.model, 'replay_supported', True) and os.environ.get('GCC_REPLAY_FORCE') != '1':
            return False            # (GCC_REPLAY_FORCE=1: timing experiments only -- a recording repeats its dropout masks)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                (torch.distributed.get_world_size() > 1 or os.environ.get('GCC_DP_FORCE_BUCKETS') == '1'):
            # data parallelism: the iteration holds gradient all-reduces.  Through the C ABI's communicator (dist.comm_route()
            # 'native') they are launch-like calls of the library and part of the recording (replayed from one host thread, in the
            # recorded order on every rank); through torch.distributed they are not: eager
            # -- an EXPERIMENTAL composition (dist.experimental(), GCC_DP_EXPERIMENTAL=1): no recording with collectives has ever
            # been replayed against a real peer
            from . import dist as gdist
            return gdist.experimental() and gdist.comm_route() == 'native'
        return True

    # ------------------------------------------------------------------------------------------------
    def step(self, batch, val_batch=None):
        if not self.usable():
            self._eager(batch, val_batch if self.arch else None)
            return 'eager'
        sig = self._signature(batch, val_batch if self.arch else None)
        if self.shapes is not None and sig != self.shapes:
            self.invalidate()
        self.shapes = sig
        a = self._stage(0, batch)
        b = self._stage(1, val_batch) if self.arch else None
        if self.rec is not None:
            self._run()
            return 'replay'
        if self.seen < self.warmup:
            self.seen += 1
            self._eager(a, b)
            return 'eager'
        self._record(a, b)
        return 'record'

    def _record(self, a, b):
        lib = ops.lib()
        torch.cuda.synchronize()                    # the pool's first allocations must not race anything pending
        self.pool = torch.cuda.MemPool()
        h = C.c_void_p()
        ops.check(lib.gcc_replay_begin(C.byref(h)), 'gcc_replay_begin')
        ops.RECORDING = True
        ops._dynamic.clear()
        ok = False
        try:
            with torch.cuda.use_mem_pool(self.pool):
                self._eager(a, b)
            ok = True
        finally:
            ops.RECORDING = False
            rc = lib.gcc_replay_end(h, self.threads)
            self.dynamic = list(ops._dynamic)
            ops._dynamic.clear()
            if not ok or rc != 0:
                lib.gcc_replay_destroy(h)
                self.pool = None
        ops.check(rc, 'gcc_replay_end')
        self.rec = h

    def _run(self):
        for obj, tag in self.dynamic:
            obj.replay_update(self.rec, tag)
        ops.check(ops.lib().gcc_replay_run(self.rec), 'gcc_replay_run')
        self.replayed += 1

    def info(self):
        if self.rec is None:
            return None
        lib = ops.lib()
        return {k: int(lib.gcc_replay_info(self.rec, i)) for i, k in enumerate(('entries', 'launches', 'streams', 'threads', 'argument_bytes'))}

    def invalidate(self):
        """forget the recording (a launch argument it holds is about to change); the next step() warms up again and records"""
        if self.rec is not None:
            torch.cuda.synchronize()
            ops.lib().gcc_replay_destroy(self.rec)
            self.rec = None
            self.pool = None
            self.dynamic = []
        self.seen = max(self.warmup - 1, 0)         # one eager iteration with the new arguments, then record

    def __del__(self):
        try:
            if self.rec is not None:
                torch.cuda.synchronize()
                ops.lib().gcc_replay_destroy(self.rec)
        except Exception:
            pass
