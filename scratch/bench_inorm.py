"""InstanceNorm: three-launch pipeline against the one-launch kernels (GCC_INORM_LPP=1|2|4 picks the slab width)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops

dev = torch.device('cuda:0')
ops.lib()


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('LPP', os.environ.get('GCC_INORM_LPP', 'default'))
for N, C, H, W in ((1, 256, 64, 64), (1, 96, 64, 64), (1, 128, 128, 128), (1, 48, 128, 128), (1, 64, 256, 256), (1, 24, 256, 256),
                   (4, 256, 64, 64), (16, 256, 64, 64)):
    x = ops.new_act(N, C, H, W, dev); x.normal_()
    y = ops.new_act(N, C, H, W, dev)
    g = ops.new_act(N, C, H, W, dev); g.normal_()
    dx = ops.new_act(N, C, H, W, dev)
    st = ops.INState(N, C, dev)

    def f3():
        ops.in_finalize(ops.channel_stats(x), H * W, st)
        ops.bnact_fwd(x, y, scale=st.scale, shift=st.shift, act=ops.ACT_RELU, groups=N)

    def f1():
        ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)

    def b3():
        ops.bnact_bwd(x, y, g, dx, bn=st, act=ops.ACT_RELU, groups=N)

    def b1():
        ops.inorm_bwd(x, y, g, dx, st, act=ops.ACT_RELU)
    print('N%d C%d %dx%d  fwd 3-launch %.1f us  1-launch %.1f us | bwd 3-launch %.1f us  1-launch %.1f us' % (
        N, C, H, W, timeit(f3), timeit(f1), timeit(b3), timeit(b1)), flush=True)
