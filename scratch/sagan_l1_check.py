import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from gcc_amd import ops
from tests.test_kernels_gpu import rb, to_dev, to_cpu, master_cl, close
DEV = 'cuda:0'
g = torch.Generator().manual_seed(5)
for (N, Cin, Cout, h, k, s, p) in [(4, 128, 64, 1, 4, 1, 0), (4, 64, 32, 4, 4, 2, 1)]:
    x = rb(torch.randn(N, Cin, h, h, generator=g))
    w = rb(torch.randn(Cin, Cout, k, k, generator=g) * 0.1)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv_transpose2d(xr, wr, None, stride=s, padding=p)
    H = y_ref.shape[2]
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    m = master_cl(w)
    wp, wtp = ops.pack_weights(m)
    xd = to_dev(x)
    y, stats = ops.conv_dgrad(xd, wtp, Cout, H, H, k, s, p, want_stats=True)
    yg = to_cpu(y)
    print('fwd rel', float((yg - y_ref.detach()).norm() / y_ref.norm()))
    st = stats.sum(0).cpu()
    print('stats sum err', float((st[0] - yg.sum((0, 2, 3))).abs().max()), 'sumsq err', float((st[1] - (yg * yg).sum((0, 2, 3))).abs().max()), 'ref mag', float((yg*yg).sum((0,2,3)).max()))
    dyd = to_dev(dy)
    dx = ops.conv_fprop(dyd, wp, Cin, k, s, p)
    print('dx rel', float((to_cpu(dx) - xr.grad).norm() / xr.grad.norm()))
    dw = torch.zeros_like(m)
    ops.conv_wgrad(dyd, xd, dw, k, s, p)
    print('dw rel', float((dw.cpu() - wr.grad).norm() / wr.grad.norm()))
