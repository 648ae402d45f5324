"""BatchNorm / activation streaming passes at the PatchGAN sizes of the headline step (N = 16) and SRGAN's discriminator at 96 -> 384: us per
call form and effective HBM rate (minimum bytes the form must move / time)"""
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
shapes = [('L2 256 @64', 16, 256, 64, 64), ('L3 512 @32', 16, 512, 32, 32), ('L4 1024 @31', 16, 1024, 31, 31), ('L1 128 @128 (gate only)', 16, 128, 128, 128),
          ('SR-D 128 @192', 16, 128, 192, 192), ('SR-D 64 @192', 16, 64, 192, 192), ('SR-D 256 @96', 16, 256, 96, 96), ('SR-D 64 @384', 16, 64, 384, 384), ('SR-D 128 @384', 16, 128, 384, 384)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
print('%-26s %8s | %18s %18s | %22s %22s %22s' % ('tensor', 'MB', 'fwd plain', 'fwd gate', 'bwd plain (5T)', 'bwd gate+dalpha (5T)', 'bwd eval-less apply'))
for name, N, Cc, H, W in shapes:
    x = ops.new_act(N, Cc, H, W, DEV); x.normal_()
    g = ops.new_act(N, Cc, H, W, DEV); g.normal_()
    y = ops.new_act(N, Cc, H, W, DEV); dx = ops.new_act(N, Cc, H, W, DEV)
    st = ops.BNState(Cc, DEV); st.mean.normal_(); st.rstd.fill_(1.0); st.scale.fill_(1.0); st.shift.normal_()
    gamma = torch.ones(Cc, device=DEV); beta = torch.zeros(Cc, device=DEV)
    dgamma = torch.zeros(Cc, device=DEV); dbeta = torch.zeros(Cc, device=DEV); dalpha = torch.zeros(Cc, device=DEV)
    mask = (torch.rand(Cc, device=DEV) > 0.3).float()
    T = N * Cc * H * W * 2 / 1e6
    f0 = med(lambda: ops.bnact_fwd(x, y, scale=st.scale, shift=st.shift, act=ops.ACT_LRELU))
    f1 = med(lambda: ops.bnact_fwd(x, y, scale=st.scale, shift=st.shift, gate=mask, act=ops.ACT_LRELU))
    b0 = med(lambda: ops.bnact_bwd(x, None, g, dx, bn=st, gamma=gamma, beta=beta, act=ops.ACT_LRELU, act2=ops.ACT_LRELU, dgamma=dgamma, dbeta=dbeta))
    b1 = med(lambda: ops.bnact_bwd(x, None, g, dx, bn=st, gamma=gamma, beta=beta, gate=mask, act=ops.ACT_LRELU, act2=ops.ACT_LRELU, dgamma=dgamma, dbeta=dbeta, dalpha=dalpha))
    b2 = med(lambda: ops.bnact_bwd(x, None, g, dx, bn=st, gamma=gamma, beta=beta, gate=mask, act=ops.ACT_LRELU, act2=ops.ACT_LRELU))
    r = lambda k, t: '%7.1f us %5.2f TB/s' % (t, k * T / t)
    print('%-26s %8.1f | %s %s | %s %s %s' % (name, T, r(2, f0), r(2, f1), r(5, b0), r(5, b1), r(5, b2)), flush=True)
