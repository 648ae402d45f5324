"""the discriminators' L4 weight gradient (512 -> 1024, k4 s1 p1 at 32 x 32, N = 16): wgrad_kernel<true> against wgrad_ts_kernel,
for the alone plan (256 workgroups) and the production plan (128)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
CASES = [(16, 32, 32, 512, 1024), (16, 32, 32, 256, 512), (8, 32, 32, 512, 1024), (16, 64, 64, 256, 512)]
for (N, H, W, Ci, Co) in CASES:
    x = torch.randn(N, Ci, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Co, H - 1, W - 1, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(Co, Ci, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * (H - 1) * (W - 1) * Co * Ci * 16
    for wgs in (256, 128):
        ops.set_plan(wgrad_wgs_big=wgs)
        for ts in (0, 1):
            lib.gcc_set_option(_lib.OPT_WGRAD_TS, ts)
            for _ in range(3):
                ops.conv_wgrad(x, dy, dw, 4, 1, 1)
            torch.cuda.synchronize()
            ts_ = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.conv_wgrad(x, dy, dw, 4, 1, 1); e1.record(); torch.cuda.synchronize()
                ts_.append(e0.elapsed_time(e1) * 1e3)
            ts_.sort()
            print('N%d %dx%d %d->%d  wgs %3d  ts %d : median %7.1f us  min %7.1f us  %7.1f TFLOP/s' % (
                N, H, W, Ci, Co, wgs, ts, ts_[10], ts_[0], fl / ts_[10] / 1e6), flush=True)
