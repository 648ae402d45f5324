"""does a streaming pass run slower right behind an MFMA-heavy kernel?  bnact_fwd on the conv's 151 MB output, timed alone (rotating
buffers) and as the second kernel of [conv 64 -> 128 k3 s1 @192 x 192 x 16 ; bnact_fwd of its output] pairs (HIP events around each)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops
dev = torch.device('cuda:0')
ops.lib()
N, Ci, Co, H = 16, 64, 128, 192
R = 3
xs = [ops.new_act(N, Ci, H, H, dev) for _ in range(R)]
for x in xs:
    x.normal_()
raws = [ops.new_act(N, Co, H, H, dev) for _ in range(R)]
acts = [ops.new_act(N, Co, H, H, dev) for _ in range(R)]
m = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
w, wt = ops.pack_weights(m)
sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev); gate = torch.ones(Co, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
for mode in ('bn only', 'conv + bn', 'conv + bn, 20 us idle between'):
    tc, tb = [], []
    for it in range(12):
        r = it % R
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        if mode != 'bn only':
            ops.conv_fprop(xs[r], w, Co, 3, 1, 1, out=raws[r])
        e1.record()
        if mode.endswith('between'):
            torch.cuda._sleep(40000)
            e1 = ev(); e1.record()
        ops.bnact_fwd(raws[r], acts[r], scale=sc, shift=sh, gate=gate, act=ops.ACT_LRELU)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            tc.append(e0.elapsed_time(e1) * 1e3); tb.append(e1.elapsed_time(e2) * 1e3)
    tc.sort(); tb.sort()
    print('%-32s conv %7.1f us   bnact_fwd %7.1f us (median of %d)' % (mode, tc[len(tc) // 2], tb[len(tb) // 2], len(tb)), flush=True)
