"""BatchNorm forward / backward passes against a plain copy of the same bytes (HBM floor of the streaming kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, C, H in ((16, 64, 128), (16, 128, 128), (16, 256, 64), (16, 512, 32), (16, 1024, 31), (16, 128, 64), (16, 256, 32)):
    x = ops.new_act(N, C, H, H, dev); x.normal_()
    y = ops.new_act(N, C, H, H, dev)
    g = ops.new_act(N, C, H, H, dev); g.normal_()
    dx = ops.new_act(N, C, H, H, dev)
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    bn = type('S', (), {})()
    bn.mean = torch.zeros(C, device=dev); bn.rstd = torch.ones(C, device=dev)
    mb = N * C * H * H * 2 / 1e6
    xb = x.permute(0, 2, 3, 1); yb = y.permute(0, 2, 3, 1)
    t_copy = timeit(lambda: yb.copy_(xb))
    t_fwd = timeit(lambda: ops.bnact_fwd(x, y, scale=sc, shift=sh, act=ops.ACT_LRELU))
    t_bwd = timeit(lambda: ops.bnact_bwd(x, None, g, dx, bn=bn, gamma=gamma, beta=beta, act=ops.ACT_LRELU, dgamma=dgamma, dbeta=dbeta))
    gate = torch.ones(C, device=dev); dalpha = torch.zeros(C, device=dev)
    t_bwdg = timeit(lambda: ops.bnact_bwd(x, None, g, dx, bn=bn, gamma=gamma, beta=beta, gate=gate, act=ops.ACT_LRELU, act2=ops.ACT_LRELU, dgamma=dgamma, dbeta=dbeta, dalpha=dalpha))
    print('N%d C%4d %3dx%-3d %5.1f MB | copy %5.1f us (%.1f TB/s) | bnact_fwd %5.1f us (%.1f TB/s) | bnact_bwd %6.1f us (3 launches; 14 B/elem alg. -> %.1f TB/s)'
          % (N, C, H, H, mb, t_copy, 2 * mb / t_copy, t_fwd, 2 * mb / t_fwd, t_bwd, 7 * mb / t_bwd) + ' | gated bwd %6.1f us' % t_bwdg, flush=True)
