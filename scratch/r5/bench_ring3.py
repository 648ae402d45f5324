"""3 x 3 stride-1 layers between <= 64-channel tensors (SRGAN's SRResNet trunk at 96 x 96, VGG19 conv1_2 at 384 x 384; N = 16): fprop with
statistics / fprop + bias + ReLU / dgrad / wgrad per launch, ring-walk route (conv_ring3.hip) against igemm_kernel (GCC_OPT_IGEMM_THIN 0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops, _lib
DEV = torch.device('cuda:0')
lib = _lib.load()
def med(fn, n=9):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
g = torch.Generator().manual_seed(0)
print('%-34s %10s %10s %10s %10s   GFLOP  HBM floor us (in + out at 5.5 TB/s)' % ('layer', 'fprop+st', 'fprop+act', 'dgrad', 'wgrad'))
for name, N, H, W, Ci, Co in (('trunk teacher 64->64 @96', 16, 96, 96, 64, 64), ('trunk student 24->24 @96', 16, 96, 96, 24, 24),
                              ('trunk student 40->48 @96', 16, 96, 96, 40, 48), ('VGG conv1_2 64->64 @384', 16, 384, 384, 64, 64),
                              ('trunk teacher @24 (24->96 cfg)', 16, 24, 24, 64, 64)):
    x = ops.new_act(N, Ci, H, W, DEV); x.normal_()
    dy = ops.new_act(N, Co, H, W, DEV); dy.normal_()
    m = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(Co, device=DEV)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, H, W, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
    dw = torch.zeros_like(m)
    fl = 2.0 * N * H * W * Co * 9 * Ci
    hbm = (N * H * W * (Ci + Co) * 2) / 5.5e12 * 1e6
    for thin in (1, 0):
        lib.gcc_set_option(_lib.OPT_IGEMM_THIN, thin)
        ts = med(lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, want_stats=True))
        ta = med(lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, bias=b, act=ops.ACT_RELU))
        td = med(lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 3, 1, 1, out=dx))
        tw = med(lambda: ops.conv_wgrad(x, dy, dw, 3, 1, 1))
        print('%-34s %10.1f %10.1f %10.1f %10.1f   %5.1f  %5.1f   %s' % (name, ts, ta, td, tw, fl / 1e9, hbm, 'ring-walk' if thin else 'igemm_kernel'), flush=True)
    lib.gcc_set_option(_lib.OPT_IGEMM_THIN, -1)
