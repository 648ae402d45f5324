"""Stream schedule shared by the model classes: the online teacher's iteration runs on its own HIP stream.

The teacher's step and the student's forward / teacher-independent loss terms have no data dependence, so they are
enqueued on two streams; the student's stream joins the teacher's where the teacher's features are first read, and the
teacher's stream is released by an event recorded at the last main-stream launch that reads its buffers.  The many small
layers of the two networks, which cannot fill 256 CUs alone, overlap; the arithmetic is unchanged.
``GCC_CONCURRENT_TEACHER=0`` (or ``model.serialize_streams = True``, used by bench.py's profiled steps) keeps everything
on one stream.

Host side (round 3): GCC_TEACHER_THREAD=1 hands the teacher's step to a second host thread.  ctypes releases the GIL inside
every library call, so the ~4.3 us `hipLaunchKernel` halves of the two threads' launches overlap while the Python halves take
turns: CycleGAN at batch 1 gains 9 % (34.7 -> 31.5 ms), SAGAN loses 15 % and SRGAN becomes erratic (14.7 / 35 ms) -- GIL
ping-pong -- so it is off by default; the launch-bound models' answer is gcc_amd.replay (the iteration re-issued from native
code).  Never under data parallelism (RCCL calls of one communicator from two threads)."""
import os
import queue
import threading

import torch

from .. import ops


class TeacherStreamMixin:
    def _ensure_plan(self):
        """state THIS model's tile plan (pair split, halo columns, weight-gradient split targets) for the launches the calling
        thread is about to enqueue: ops.set_plan -- thread-local host state that travels with every convolution call
        (gcc_conv_t.plan); the library itself holds no plan any more (round 5).  Explicit GCC_* environment values win (ops)."""
        plan = getattr(self, '_plan', None)
        if plan is None:
            return
        pair, wgs_big, wgs = plan[:3]
        ops.set_plan(pair=max(pair, 0), halo_hc=1 if pair == 1 else 0, wgrad_wgs_big=max(wgs_big, 0), wgrad_wgs=max(wgs, 0))

    def restore_library_plan(self):
        """back to the library's default plan on this thread (model teardown / tests that go on to call the kernels directly)"""
        self._plan = None
        ops.set_plan()

    def set_stream_schedule(self, concurrent, plan=None):
        """concurrent=True: the production schedule (student, online teacher, auxiliary and weight-gradient streams).
        False: every launch on one stream.  plan: the library's tile plan -- 'production' (what the multi-stream schedule runs:
        no pair split, half-chip weight-gradient splits) or 'alone' (for launches that have the chip to themselves:
        the pair split, full-chip weight-gradient splits); default: 'production' with concurrent streams, 'alone' without.
        bench.py times one single-stream step under each plan (a launch's duration is then the kernel's own)."""
        from .. import engine
        self.serialize_streams = not concurrent
        if getattr(self, 'teacher_model', None) is not None:
            self.teacher_model.serialize_streams = not concurrent
        engine.OVERLAP_WGRAD = bool(concurrent) and engine.overlap_wgrad_default(getattr(self, '_world', 1))
        if plan is None:
            plan = 'production' if concurrent else 'alone'
        assert plan in ('production', 'alone')
        # GCC_PAIR_CONCURRENT=1: keep the pair split in the production plan too (A/B hook)
        pair_prod = 1 if os.environ.get('GCC_PAIR_CONCURRENT', '0') == '1' else 0
        # Workgroup targets of the split weight-gradient launches: the library's defaults (256 / 512) are for a launch that has
        # the chip to itself; on the weight-gradient side stream of the production schedule half of that is faster end to end
        # (+0.6 %, profiles/r02_ab_tables.md r03w / r03x; re-measured with this round's kernels: profiles/r3l_ab_plans.txt)
        self._plan = (pair_prod, 128, 256) if plan == 'production' else (1, -1, -1)
        if getattr(self, 'teacher_model', None) is not None:
            self.teacher_model._plan = self._plan
        self._ensure_plan()

    def _teacher_stream(self):
        self._ensure_plan()
        if getattr(self, 'serialize_streams', False):
            return False
        if getattr(self, '_tstream', None) is None:
            on = os.environ.get('GCC_CONCURRENT_TEACHER', '1') != '0'
            self._tstream = torch.cuda.Stream(device=self.device) if on else False
            if on and getattr(self, '_plan', None) is None:
                self._plan = (1 if os.environ.get('GCC_PAIR_CONCURRENT', '0') == '1' else 0, 128, 256)
                self._ensure_plan()
        return self._tstream

    def _mark_teacher_free(self):
        """main stream: no later launch reads the teacher's buffers -- its stream may move on from here"""
        ev = getattr(self, '_teacher_free', None)
        if ev is None:
            ev = self._teacher_free = ops.Event()      # re-recorded: the teacher's wait holds the record it saw
        ev.record()

    def _note_input(self, input):
        """set_input(): order this stream -- and, through _release_teacher_stream, the teacher's -- behind whatever produced
        a device-resident batch.  A batch dict may carry 'ready' (a torch.cuda.Event recorded by its producer after the last
        kernel that wrote its tensors: gcc_amd.data's loaders and bench.py set it); without one, device tensors are assumed
        to have been produced on the stream set_input is called on, and an event recorded here stands for them (the
        teacher then also waits for everything else already enqueued on this stream).  Host tensors need nothing: the
        H2D copy is issued on the consuming stream itself.  The tensors are marked as used on this stream so that the
        caching allocator does not hand their memory to the producer's stream while the copies below are still pending."""
        ev = input.get('ready') if hasattr(input, 'get') else None
        dev = [v for v in input.values() if torch.is_tensor(v) and v.is_cuda] if hasattr(input, 'values') else []
        if dev:
            cur = ops.current_stream()
            if ev is not None:
                ops.wait_event(cur, ev)
            else:
                ev = getattr(self, '_own_input_event', None)
                if ev is None:
                    ev = self._own_input_event = ops.Event()
                ev.record(cur)
            for t in dev:
                t.record_stream(cur)
        self._input_ready = ev if dev else None

    def _release_teacher_stream(self, ts, wait_free=True):
        """wait_free=False: the work about to be enqueued touches nothing the student still reads (Pix2Pix ARCH_EARLY)"""
        ev = getattr(self, '_teacher_free', None)
        if not wait_free:
            pass
        elif ev is not None:
            ops.wait_event(ts, ev)
        else:
            ops.wait_stream(ts, ops.current_stream())
        # the batch the teacher is about to read (set_input on its own stream) must exist: ADVICE r1, race on a
        # device-resident batch written by main-stream kernels after _teacher_free was recorded
        ready = getattr(self, '_input_ready', None)
        if ready is not None:
            ops.wait_event(ts, ready)

    teacher_thread = False        # class default: enqueue the teacher's step from the calling thread

    def _teacher_thread_on(self):
        env = os.environ.get('GCC_TEACHER_THREAD')
        on = self.teacher_thread if env is None else env == '1'
        if ops.RECORDING:               # a recording is the calling thread's: everything is enqueued from it
            return False
        if on and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            on = False
        return on

    def _run_teacher(self, fn):
        """run fn() (teacher work) on the teacher's stream if there is one; returns the stream (or False).  With the
        enqueue thread fn runs there: nothing it writes on the host (the teacher's contexts, target features) may be read
        before _join(ts)."""
        ts = self._teacher_stream()
        if ts:
            self._release_teacher_stream(ts)
            if self._teacher_thread_on():
                _EnqueueThread.get(self.device).submit(fn, ts)
            else:
                with ops.on_stream(ts):
                    fn()
        else:
            fn()
        return ts

    def _aux_stream(self):
        """second stream of the student: work that does not depend on the generator's output (the discriminator's pass
        over the real pair) runs here next to the generator's forward.  GCC_EARLY_DREAL=0 turns it off."""
        if getattr(self, 'serialize_streams', False) or not torch.cuda.is_available():
            return False
        if getattr(self, '_astream', None) is None:
            on = os.environ.get('GCC_EARLY_DREAL', '1') != '0' and os.environ.get('GCC_CONCURRENT_TEACHER', '1') != '0'
            self._astream = torch.cuda.Stream(device=self.device) if on else False
        return self._astream

    @staticmethod
    def _join(ts):
        if ts:
            _EnqueueThread.drain()           # the teacher's launches must all be enqueued before the stream is waited for
            ops.wait_stream(ops.current_stream(), ts)


class _EnqueueThread:
    """one worker per device that enqueues the teacher's launches on the teacher's stream; drain() returns once everything
    submitted has been enqueued and re-raises what the step raised"""
    _inst = {}

    def __init__(self, device):
        self.device = device
        self.q = queue.SimpleQueue()
        self.done = threading.Semaphore(0)
        self.pending = 0
        self.error = None
        self.thread = threading.Thread(target=self._loop, name='gcc-teacher-enqueue', daemon=True)
        self.thread.start()

    @classmethod
    def get(cls, device):
        key = str(device)
        inst = cls._inst.get(key)
        if inst is None:
            inst = cls._inst[key] = _EnqueueThread(device)
        return inst

    def _loop(self):
        torch.cuda.set_device(self.device)
        while True:
            fn, ts, plan = self.q.get()
            try:
                ops.set_plan(**plan)             # the submitting thread's tile plan (thread-local state of ops)
                with ops.on_stream(ts):
                    fn()
            except BaseException as e:          # handed to the submitting thread at drain()
                self.error = e
            self.done.release()

    def submit(self, fn, ts):
        self._wait()                            # one step at a time: the steps of a model share its host state
        self.pending += 1
        self.q.put((fn, ts, ops.current_plan()))

    def _wait(self):
        while self.pending:
            self.done.acquire()
            self.pending -= 1
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    @classmethod
    def drain(cls):
        for inst in cls._inst.values():
            inst._wait()
