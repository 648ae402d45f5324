"""HBM efficiency of the BatchNorm/activation streaming kernels on the PatchGAN shapes (N=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops
DEV = 'cuda:0'

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for name, N, C, H, W, gate in (('D.L1 gate-after-act', 16, 128, 128, 128, True), ('D.L2', 16, 256, 64, 64, True), ('D.L3', 16, 512, 32, 32, True),
                               ('D.L4', 16, 1024, 31, 31, True), ('D.L2 plain', 16, 256, 64, 64, False), ('tG.e1', 16, 128, 64, 64, False)):
    x = ops.new_act(N, C, H, W, DEV); x.copy_(torch.randn(N, C, H, W).bfloat16().to(DEV))
    y = ops.new_act(N, C, H, W, DEV); g = ops.new_act(N, C, H, W, DEV); g.copy_(torch.randn(N, C, H, W).bfloat16().to(DEV))
    dx = ops.new_act(N, C, H, W, DEV)
    st = ops.BNState(C, DEV); st.rstd.fill_(1.0); st.scale.fill_(1.0)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    mask = torch.ones(C, device=DEV) if gate else None
    dg, db, da = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    elems = N * C * H * W
    if name.startswith('D.L1'):
        tf = timeit(lambda: ops.bnact_fwd(x, y, gate=mask, gate_after_act=True))
        tb = timeit(lambda: ops.bnact_bwd(x, None, g, dx, gate=mask, gate_after_act=True, in_act=ops.ACT_LRELU, dalpha=da))
        print('%-22s fwd %6.1f us (%5.2f TB/s @4B/elem)   bwd %6.1f us (%5.2f TB/s @6B/elem)' % (name, tf * 1e6, elems * 4 / tf / 1e12, tb * 1e6, elems * 6 / tb / 1e12))
    else:
        tf = timeit(lambda: ops.bnact_fwd(x, y, scale=st.scale, shift=st.shift, gate=mask, act=ops.ACT_LRELU))
        tb = timeit(lambda: ops.bnact_bwd(x, None, g, dx, bn=st, gamma=gamma, beta=beta, gate=mask, act=ops.ACT_LRELU, dgamma=dg, dbeta=db, dalpha=da if gate else None))
        print('%-22s fwd %6.1f us (%5.2f TB/s @4B/elem)   bwd %6.1f us (%5.2f TB/s @12B/elem)' % (name, tf * 1e6, elems * 4 / tf / 1e12, tb * 1e6, elems * 14 / tb / 1e12))
