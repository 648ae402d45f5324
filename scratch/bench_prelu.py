"""PReLU backward (scalar slope gradient) on SRGAN's trunk tensors: us per call (rotating buffers)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()
for N, C, H in ((16, 64, 96), (16, 24, 96), (16, 64, 24), (16, 64, 192)):
    R = 6
    sets = []
    for r in range(R):
        x = ops.new_act(N, C, H, H, dev); x.normal_()
        g = ops.new_act(N, C, H, H, dev); g.normal_()
        dx = ops.new_act(N, C, H, H, dev)
        sets.append((x, g, dx))
    slope = torch.full((1,), 0.25, device=dev); ds = torch.zeros(1, device=dev)
    for s in sets:
        ops.prelu_bwd(s[0], slope, s[1], s[2], ds)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30):
        s = sets[i % R]
        ops.prelu_bwd(s[0], slope, s[1], s[2], ds)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e3
    mb = N * C * H * H * 2 / 1e6
    print('N%d C%3d %3dx%-3d %6.1f MB: prelu backward %6.1f us  %4.2f TB/s over 3 T' % (N, C, H, H, mb, t, 3 * mb / t), flush=True)
