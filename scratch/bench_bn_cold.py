"""BatchNorm / activation streaming passes on the big tensors of the SRGAN 96 -> 384 discriminator, WARM (the same buffers every call:
part of them stay in the 256 MB Infinity Cache) against COLD (R rotating buffer sets, > 1 GB in all): us per call and the rate over the
algorithmic bytes (forward 2 T, backward 5 T), beside torch's plain copy of the same tensor (2 T)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()


def timeit(fns, n=24):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, C, H in ((16, 128, 192), (16, 64, 192), (16, 64, 384), (16, 128, 96), (16, 256, 64), (16, 64, 96)):
    mb = N * C * H * H * 2 / 1e6
    R = max(2, int(1400 / (4 * mb)))
    sets = []
    for r in range(R):
        x = ops.new_act(N, C, H, H, dev); x.normal_()
        y = ops.new_act(N, C, H, H, dev)
        g = ops.new_act(N, C, H, H, dev); g.normal_()
        dx = ops.new_act(N, C, H, H, dev)
        sets.append((x, y, g, dx))
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    bn = type('S', (), {})()
    bn.mean = torch.zeros(C, device=dev); bn.rstd = torch.ones(C, device=dev)
    gate = torch.ones(C, device=dev); dalpha = torch.zeros(C, device=dev)
    out = []
    for label, use in (('warm', sets[:1]), ('cold', sets)):
        t_copy = timeit([(lambda s=s: s[1].permute(0, 2, 3, 1).copy_(s[0].permute(0, 2, 3, 1))) for s in use])
        t_fwd = timeit([(lambda s=s: ops.bnact_fwd(s[0], s[1], scale=sc, shift=sh, gate=gate, act=ops.ACT_LRELU)) for s in use])
        t_bwd = timeit([(lambda s=s: ops.bnact_bwd(s[0], None, s[2], s[3], bn=bn, gamma=gamma, beta=beta, gate=gate, act=ops.ACT_LRELU,
                                                   dgamma=dgamma, dbeta=dbeta, dalpha=dalpha)) for s in use])
        t_bwdy = timeit([(lambda s=s: ops.bnact_bwd(s[0], s[1], s[2], s[3], bn=bn, gamma=gamma, beta=beta, gate=gate, act=ops.ACT_LRELU,
                                                    dgamma=dgamma, dbeta=dbeta, dalpha=dalpha)) for s in use])
        t_act = timeit([(lambda s=s: ops.bnact_bwd(s[0], s[1], s[2], s[3], gate=gate, act=ops.ACT_LRELU, dalpha=dalpha)) for s in use])
        out.append('%s: copy %6.1f us %4.2f TB/s | fwd %6.1f us %4.2f TB/s | bwd %6.1f us %4.2f TB/s | bwd with saved y %6.1f us | gate-only bwd (no BN, saved y) %6.1f us' % (
            label, t_copy, 2 * mb / t_copy, t_fwd, 2 * mb / t_fwd, t_bwd, 5 * mb / t_bwd, t_bwdy, t_act))
    print('N%d C%3d %3dx%-3d %6.1f MB x %d sets | %s || %s' % (N, C, H, H, mb, R, out[0], out[1]), flush=True)
    del sets
    torch.cuda.empty_cache()
