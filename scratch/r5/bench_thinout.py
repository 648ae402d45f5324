"""the 9 x 9 thin-output layer of SRGAN at 96 -> 384 (N = 16, 384 x 384): fprop / dgrad / wgrad per launch, thin-output routes against
the generic kernels (GCC_IGEMM_THIN=0 in the environment of a second run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops
DEV = torch.device('cuda:0')
def med(fn, n=7):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
g = torch.Generator().manual_seed(0)
for Ci in (64, 24):
    N, H, W, Co, k = 16, 384, 384, 3, 9
    x = ops.new_act(N, Ci, H, W, DEV); x.normal_()
    dy = ops.new_act(N, Co, H, W, DEV); dy[:, :Co].normal_()
    m = (torch.randn(Co, Ci, k, k, generator=g) * 0.02).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, H, W, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
    dw = torch.zeros_like(m)
    fl = 2.0 * N * H * W * Co * k * k * Ci
    tf = med(lambda: ops.conv_fprop(x, w, Co, k, 1, 4, out=y, act=ops.ACT_TANH))
    td = med(lambda: ops.conv_dgrad(dy, wt, Ci, H, W, k, 1, 4, out=dx))
    tw = med(lambda: ops.conv_wgrad(x, dy, dw, k, 1, 4))
    hbm = (N * H * W * (Ci + 8) * 2) / 5.5e12 * 1e6
    print('Ci %2d (GCC_IGEMM_THIN=%s): fprop %7.1f us  dgrad %7.1f us  wgrad %7.1f us   (%.1f useful GFLOP each; HBM floor ~%.0f us)' % (
        Ci, os.environ.get('GCC_IGEMM_THIN', '1'), tf, td, tw, fl / 1e9, hbm), flush=True)
