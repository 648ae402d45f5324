"""3 x 3 stride-2 layers of SRGAN's discriminators at 96 -> 384 (N = 16): forward + LeakyReLU / forward with statistics / data gradient per launch,
the stride-2 halo form (GCC_OPT_IGEMM_HALO 3) against igemm_kernel (0)"""
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
print('%-28s %10s %10s %10s   GFLOP  bytes-once us at 4.5 TB/s' % ('layer (input size)', 'fprop+act', 'fprop+st', 'dgrad'))
for name, N, H, W, Ci, Co in (('128->128 @384', 16, 384, 384, 128, 128), ('256->256 @192', 16, 192, 192, 256, 256), ('128->128 @192', 16, 192, 192, 128, 128),
                              ('512->512 @96', 16, 96, 96, 512, 512), ('64->64 @384 (not routed)', 16, 384, 384, 64, 64)):
    x = ops.new_act(N, Ci, H, W, DEV); x.normal_()
    dy = ops.new_act(N, Co, H // 2, W // 2, DEV); dy.normal_()
    m = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(Co, device=DEV)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, H // 2, W // 2, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
    fl = 2.0 * N * (H // 2) * (W // 2) * Co * 9 * Ci
    hbm = (N * H * W * Ci + N * (H // 2) * (W // 2) * Co) * 2 / 4.5e12 * 1e6
    for halo in (3, 0):
        lib.gcc_set_option(_lib.OPT_IGEMM_HALO, halo)
        ta = med(lambda: ops.conv_fprop(x, w, Co, 3, 2, 1, out=y, bias=b, act=ops.ACT_LRELU, slope=0.2))
        ts = med(lambda: ops.conv_fprop(x, w, Co, 3, 2, 1, out=y, want_stats=True))
        td = med(lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 3, 2, 1, out=dx))
        print('%-28s %10.1f %10.1f %10.1f   %5.1f  %5.1f   %s' % (name, ta, ts, td, fl / 1e9, hbm, 'halo' if halo else 'igemm_kernel'), flush=True)
    lib.gcc_set_option(_lib.OPT_IGEMM_HALO, -1)
