"""plain training-mode BatchNorm + LeakyReLU backward (y given): the one-launch grid kernel (gcc_bn_bwd_one_launch) against reduce + finalize + apply,
by tensor size (N = 16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops
DEV = torch.device('cuda:0')
def med(fn, n=15):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
shapes = [(16, 64, 24, 24), (16, 64, 48, 48), (16, 64, 96, 96), (16, 128, 96, 96), (16, 256, 96, 96), (16, 64, 192, 192), (16, 128, 192, 192), (16, 512, 48, 48), (16, 512, 24, 24), (16, 256, 64, 64), (16, 1024, 31, 31)]
print('%-22s %8s %12s %12s' % ('tensor', 'MB', 'one launch', '3 launches'))
for N, Cc, H, W in shapes:
    x = ops.new_act(N, Cc, H, W, DEV); x.normal_()
    g = ops.new_act(N, Cc, H, W, DEV); g.normal_()
    y = ops.new_act(N, Cc, H, W, DEV); y.normal_()
    dx = ops.new_act(N, Cc, H, W, DEV)
    st = ops.BNState(Cc, DEV); st.mean.normal_(); st.rstd.fill_(1.0); st.scale.fill_(1.0); st.shift.normal_()
    gamma = torch.ones(Cc, device=DEV); beta = torch.zeros(Cc, device=DEV)
    dgamma = torch.zeros(Cc, device=DEV); dbeta = torch.zeros(Cc, device=DEV)
    fn = lambda: ops.bnact_bwd(x, y, g, dx, bn=st, gamma=gamma, beta=beta, act=ops.ACT_LRELU, dgamma=dgamma, dbeta=dbeta)
    ops.BN_BWD_GRID = True; t1 = med(fn)
    ops.BN_BWD_GRID = False; t3 = med(fn)
    ops.BN_BWD_GRID = True
    print('%-22s %8.1f %9.1f us %9.1f us' % ('%d x %d x %d' % (Cc, H, W), N * Cc * H * W * 2 / 1e6, t1, t3), flush=True)
