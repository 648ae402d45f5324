"""Stream schedule shared by the model classes: the online teacher's iteration runs on its own HIP stream.

The teacher's step and the student's forward / teacher-independent loss terms have no data dependence, so they are
enqueued on two streams; the student's stream joins the teacher's where the teacher's features are first read, and the
teacher's stream is released by an event recorded at the last main-stream launch that reads its buffers.  The many small
layers of the two networks, which cannot fill 256 CUs alone, overlap; the arithmetic is unchanged.
``GCC_CONCURRENT_TEACHER=0`` (or ``model.serialize_streams = True``, used by bench.py's profiled steps) keeps everything
on one stream."""
import os

import torch

from .. import ops


class TeacherStreamMixin:
    def set_stream_schedule(self, concurrent, plan=None):
        """concurrent=True: the production schedule (student, online teacher, auxiliary and weight-gradient streams).
        False: every launch on one stream.  plan: the library's tile plan -- 'production' (what the multi-stream schedule runs:
        no pair split, half-chip weight-gradient splits) or 'alone' (for launches that have the chip to themselves:
        GCC_OPT_IGEMM_PAIR, full-chip weight-gradient splits); default: 'production' with concurrent streams, 'alone' without.
        bench.py times one single-stream step under each plan (a launch's duration is then the kernel's own)."""
        from .. import _lib, engine
        self.serialize_streams = not concurrent
        if getattr(self, 'teacher_model', None) is not None:
            self.teacher_model.serialize_streams = not concurrent
        engine.OVERLAP_WGRAD = bool(concurrent) and os.environ.get('GCC_OVERLAP_WGRAD', '1') != '0'
        if plan is None:
            plan = 'production' if concurrent else 'alone'
        assert plan in ('production', 'alone')
        # GCC_PAIR_CONCURRENT=1: keep the pair split in the production plan too (A/B hook)
        pair_prod = 1 if os.environ.get('GCC_PAIR_CONCURRENT', '0') == '1' else 0
        ops.lib().gcc_set_option(_lib.OPT_IGEMM_PAIR, pair_prod if plan == 'production' else 1)
        self._apply_wgrad_plan(plan == 'production')

    @staticmethod
    def _apply_wgrad_plan(concurrent):
        """Workgroup targets of the split weight-gradient launches.  The library's defaults (256 / 512) are for a launch that
        has the chip to itself; on the weight-gradient side stream of the production schedule half of that is faster end to
        end (+0.6 %, profiles/r02_ab_tables.md r03w / r03x): half-chip launches beside the main stream's chain, half the fp32
        slab traffic -- although the same launches alone run at 365 instead of 538 TFLOP/s.  An explicit GCC_WGRAD_WGS* wins."""
        from .. import _lib
        lib = ops.lib()
        for opt_id, env, side in ((_lib.OPT_WGRAD_WGS_BIG, 'GCC_WGRAD_WGS_BIG', 128), (_lib.OPT_WGRAD_WGS, 'GCC_WGRAD_WGS', 256)):
            if env not in os.environ:
                lib.gcc_set_option(opt_id, side if concurrent else -1)

    def _teacher_stream(self):
        if getattr(self, 'serialize_streams', False):
            return False
        if getattr(self, '_tstream', None) is None:
            on = os.environ.get('GCC_CONCURRENT_TEACHER', '1') != '0'
            self._tstream = torch.cuda.Stream(device=self.device) if on else False
            if on:
                self._apply_wgrad_plan(True)
        return self._tstream

    def _mark_teacher_free(self):
        """main stream: no later launch reads the teacher's buffers -- its stream may move on from here"""
        self._teacher_free = torch.cuda.Event()
        self._teacher_free.record(ops.current_stream())

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
                cur.wait_event(ev)
            else:
                ev = torch.cuda.Event()
                ev.record(cur)
            for t in dev:
                t.record_stream(cur)
        self._input_ready = ev if dev else None

    def _release_teacher_stream(self, ts):
        ev = getattr(self, '_teacher_free', None)
        if ev is not None:
            ts.wait_event(ev)
        else:
            ts.wait_stream(ops.current_stream())
        # the batch the teacher is about to read (set_input on its own stream) must exist: ADVICE r1, race on a
        # device-resident batch written by main-stream kernels after _teacher_free was recorded
        ready = getattr(self, '_input_ready', None)
        if ready is not None:
            ts.wait_event(ready)

    def _run_teacher(self, fn):
        """run fn() (teacher work) on the teacher's stream if there is one; returns the stream (or False)"""
        ts = self._teacher_stream()
        if ts:
            self._release_teacher_stream(ts)
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
            ops.current_stream().wait_stream(ts)
