import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from gcc_amd import ops
from tests.test_kernels_gpu import rb, to_dev, to_cpu, master_cl
g = torch.Generator().manual_seed(5)
N, Cin, Cout, h, k, s, p = 4, 64, 32, 4, 4, 2, 1
x = rb(torch.randn(N, Cin, h, h, generator=g))
w = rb(torch.randn(Cin, Cout, k, k, generator=g) * 0.1)
b = torch.randn(Cout, generator=g)
y_ref = F.conv_transpose2d(x, w, b, stride=s, padding=p)
m = master_cl(w)
wp, wtp = ops.pack_weights(m)
xd = to_dev(x)
out = ops.new_act(N, Cout, 8, 8, 'cuda:0')
y, stats = ops.conv_dgrad(xd, wtp, Cout, 8, 8, k, s, p, out=out, bias=b.cuda(), want_stats=True)
yg = to_cpu(y)
print('fwd rel', float((yg - y_ref).norm() / y_ref.norm()), stats.shape)
st = stats.sum(0).cpu()
print('stats sum err', float((st[0] - yg.sum((0, 2, 3))).abs().max()), 'sumsq err', float((st[1] - (yg * yg).sum((0, 2, 3))).abs().max()))
print(st[0][:6], yg.sum((0,2,3))[:6])
