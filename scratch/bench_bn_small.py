"""BatchNorm forward / backward on the small inner U-Net maps (launch-latency regime)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, C, H in ((16, 512, 1), (16, 512, 2), (16, 512, 4), (16, 512, 8), (16, 256, 16), (16, 128, 32), (16, 1024, 31)):
    x = ops.new_act(N, C, H, H, dev); x.normal_()
    y = ops.new_act(N, C, H, H, dev)
    g = ops.new_act(N, C, H, H, dev); g.normal_()
    dx = ops.new_act(N, C, H, H, dev)
    bn = ops.BNState(C, dev) if hasattr(ops, 'BNState') else None
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    st = ops.INState(1, C, dev)
    ops.in_finalize(ops.channel_stats(x.reshape(1, C, N * H, H) if False else x), N * H * H, st) if False else None
    bn = type('S', (), {})()
    bn.mean = torch.zeros(C, device=dev); bn.rstd = torch.ones(C, device=dev)

    def bwd():
        ops.bnact_bwd(x, y, g, dx, bn=bn, gamma=gamma, beta=beta, act=ops.ACT_RELU, dgamma=dgamma, dbeta=dbeta)
    print('N%d C%d %dx%d  bnact_bwd (3 launches) %.1f us' % (N, C, H, H, timeit(bwd)), flush=True)
