"""image-layer forward (thin_fprop_kernel): k4 s2 p1, N = 16 at 256 x 256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops
DEV = torch.device('cuda:0')
def med(fn, n=21):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
g = torch.Generator().manual_seed(0)
for rep in range(2):
    for Ci, Co in ((6, 128), (3, 128), (3, 64), (3, 32)):
        N, H, W = 16, 256, 256
        x = ops.new_act(N, Ci, H, W, DEV); x[:, :Ci].normal_()
        m = (torch.randn(Co, Ci, 4, 4, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
        w, wt = ops.pack_weights(m)
        y = ops.new_act(N, Co, H // 2, W // 2, DEV)
        t = med(lambda: ops.conv_fprop(x, w, Co, 4, 2, 1, out=y, act=ops.ACT_LRELU, slope=0.2))
        print('%d -> %3d: %6.1f / %6.1f us   (out %.1f MB)' % (Ci, Co, t[0], t[1], N * H * W // 4 * Co * 2 / 1e6), flush=True)
