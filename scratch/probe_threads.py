"""Probe: enqueue the online teacher's step from a second host thread (ctypes releases the GIL inside hipLaunchKernel).
python scratch/probe_threads.py cyclegan|sagan|srgan"""
import os
import sys
import threading
import queue

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd.models import _streams

_q, _done = queue.Queue(), queue.Queue()


def _worker():
    torch.cuda.set_device(0)
    while True:
        fn, ts = _q.get()
        with torch.cuda.stream(ts):
            fn()
        _done.put(1)


threading.Thread(target=_worker, daemon=True).start()
_orig_join = _streams.TeacherStreamMixin._join
_pending = [0]


def _run_teacher(self, fn):
    ts = self._teacher_stream()
    if ts:
        self._release_teacher_stream(ts)
        _pending[0] += 1
        _q.put((fn, ts))
    else:
        fn()
    return ts


def _join(ts):
    while _pending[0]:
        _done.get()
        _pending[0] -= 1
    if ts:
        torch.cuda.current_stream().wait_stream(ts)


if os.environ.get('THREADS', '1') == '1':
    _streams.TeacherStreamMixin._run_teacher = _run_teacher
    _streams.TeacherStreamMixin._join = staticmethod(_join)
sys.argv = ['bench_models.py'] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_models.py'), run_name='__main__')
