"""python scratch/inorm_bench.py: the one-launch InstanceNorm, slab form against grid form, on CycleGAN's planes at batch 1 / 2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
cases = [(1, 256, 64, 64), (2, 256, 64, 64), (1, 96, 64, 64), (2, 96, 64, 64), (1, 128, 128, 128), (1, 48, 128, 128), (1, 64, 256, 256),
         (2, 24, 256, 256), (1, 512, 32, 32), (1, 128, 128, 128)]
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('%-22s %10s %10s %10s %10s   (us per launch, back to back)' % ('N C H W', 'fwd slab', 'fwd grid', 'bwd slab', 'bwd grid'))
for N, C, H, W in cases:
    x = ops.new_act(N, C, H, W, dev); x.normal_()
    y = ops.new_act(N, C, H, W, dev); g = ops.new_act(N, C, H, W, dev); g.normal_(); dx = ops.new_act(N, C, H, W, dev)
    st = ops.INState(N, C, dev)
    r = []
    for bwd in (0, 1):
        for grid in (0, 1):
            lib.gcc_set_option(_lib.OPT_INORM_GRID, grid)
            if bwd: r.append(timeit(lambda: ops.inorm_bwd(x, y, g, dx, st, act=ops.ACT_RELU)))
            else: r.append(timeit(lambda: ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)))
    lib.gcc_set_option(_lib.OPT_INORM_GRID, -1)
    print('%-22s %10.1f %10.1f %10.1f %10.1f' % ('%d %d %d %d' % (N, C, H, W), *r))
