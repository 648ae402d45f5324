import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
lib.gcc_diag_set(32)      # diagnostic build only: run with GCC_HIP_LIB=gcc_amd/libgcc_hip_diag.so
for N, C, H, W in [(1, 256, 64, 64), (1, 512, 32, 32), (1, 64, 256, 256)]:
    x = ops.new_act(N, C, H, W, dev); x.normal_()
    y = ops.new_act(N, C, H, W, dev); st = ops.INState(N, C, dev)
    ws = ops.inorm_workspace(dev)
    for bwd in (0, 1):
        for _ in range(5):
            if bwd: ops.inorm_bwd(x, y, y, y, st, act=ops.ACT_RELU)
            else: ops.inorm_fwd(x, y, st, act=ops.ACT_RELU)
        torch.cuda.synchronize()
        clk = ws[-16384:].view(torch.int64).view(-1, 8).cpu()
        S = int((clk[:, 0] != 0).sum())
        c = clk[:S].double()
        t0 = c[:, 0].min()
        r = (c[:, :6] - t0) * 0.01      # us
        print('%s %s: S=%d  start %.1f..%.1f | statistics pass done %.1f..%.1f | partials exchanged and folded %.1f..%.1f | end %.1f..%.1f  (us from the first workgroup\'s start; min..max over the image-0 / group-0 domain)' % (
            (N, C, H, W), 'bwd' if bwd else 'fwd', S, r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max(), r[:, 3].min(), r[:, 3].max(),
            r[:, 5].min(), r[:, 5].max()))
        ws[-16384:].zero_()
