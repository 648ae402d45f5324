"""k3 s1 p1 weight gradients of the SRGAN 96 -> 384 step (N = 16): wgrad_kernel (GCC_OPT_WGRAD_TS 0) against wgrad_ts_kernel<3>
(1: the library's plan; 2: forced), us per call (kernel + slab fold)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
CASES = [(16, 96, 96, 64, 64), (16, 192, 192, 64, 128), (16, 192, 192, 128, 256), (16, 96, 96, 256, 512), (16, 192, 192, 64, 256),
         (16, 96, 96, 64, 256), (16, 48, 48, 512, 512)]
for (N, H, W, Ci, Co) in CASES:
    x = torch.randn(N, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Co, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(Co, Ci, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * Co * Ci * 9
    for wgs in ((-1, 128, 64) if (Ci, Co) == (64, 64) else (-1,)):
        ops.set_plan(wgrad_wgs_big=wgs) if wgs > 0 else ops.set_plan()
        for ts in (0, 1):
            lib.gcc_set_option(_lib.OPT_WGRAD_TS, ts)
            for _ in range(3):
                ops.conv_wgrad(x, dy, dw, 3, 1, 1)
            torch.cuda.synchronize()
            ts_ = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.conv_wgrad(x, dy, dw, 3, 1, 1); e1.record(); torch.cuda.synchronize()
                ts_.append(e0.elapsed_time(e1) * 1e3)
            ts_.sort()
            print('N%d %dx%d %d->%d  wgs %3d  ts %d : median %7.1f us  min %7.1f us  %7.1f TFLOP/s' % (
                N, H, W, Ci, Co, wgs, ts, ts_[10], ts_[0], fl / ts_[10] / 1e6), flush=True)
    lib.gcc_set_option(_lib.OPT_WGRAD_TS, -1)
    ops.set_plan()
