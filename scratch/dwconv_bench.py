"""python scratch/dwconv_bench.py: depthwise 3x3 (reflect) forward / backward-data / backward-weight on CycleGAN's planes, us per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, C, H, W in [(1, 96, 64, 64), (2, 96, 64, 64), (1, 256, 64, 64), (2, 256, 64, 64), (1, 48, 128, 128)]:
    x = ops.new_act(N, C, H, W, dev); x.normal_()
    dy = ops.new_act(N, C, H, W, dev); dy.normal_()
    out = ops.new_act(N, C, H, W, dev)
    w = torch.randn(C, 1, 3, 3, device=dev); b = torch.randn(C, device=dev)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    print('%-18s fwd %6.1f  bwd-data %6.1f  bwd-weight (+fold) %6.1f' % ('%d %d %d %d' % (N, C, H, W), timeit(lambda: ops.dwconv_fwd(x, w, b, out)),
          timeit(lambda: ops.dwconv_bwd_data(dy, w, out)), timeit(lambda: ops.dwconv_wgrad(x, dy, dw, db))))
