"""the first PatchGAN layer's forward (thin_fprop_kernel, 6 -> 128, k4 s2 p1, N = 16 at 256 x 256) as the step runs it: one output
(teacher) / activation + gated copy (masked student), back to back on the same buffers against behind a 1 GB stream of other
traffic (cold L2 / Infinity Cache: the state the step leaves it in), with and without HIP-event brackets around every launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops
DEV = torch.device('cuda:0')
N, H, W, Ci, Co = 16, 256, 256, 6, 128
g = torch.Generator().manual_seed(0)
x = ops.new_act(N, Ci, H, W, DEV); x[:, :Ci].normal_()
m = (torch.randn(Co, Ci, 4, 4, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
b = torch.zeros(Co, device=DEV)
w, wt = ops.pack_weights(m)
y = ops.new_act(N, Co, H // 2, W // 2, DEV)
y2 = ops.new_act(N, Co, H // 2, W // 2, DEV)
gate = (torch.randint(0, 3, (Co,), generator=g).float() * 0.5).to(DEV)
junk = torch.empty(256 << 20, dtype=torch.float32, device=DEV)       # 1 GB


def run(two, cold, reps=15):
    ts = []
    for _ in range(reps + 3):
        if cold:
            junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if two:
            ops.conv_fprop(x, w, Co, 4, 2, 1, out=y, bias=b, act=ops.ACT_LRELU, slope=0.2, y2=y2, y2_mode=ops.Y2_GATE, y2_gate=gate)
        else:
            ops.conv_fprop(x, w, Co, 4, 2, 1, out=y, bias=b, act=ops.ACT_LRELU, slope=0.2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[3:])
    return ts[len(ts) // 2], ts[0]


for two in (False, True):
    for cold in (False, True):
        t = run(two, cold)
        mb = (N * H * W * 8 * 2 + N * H * W // 4 * Co * 2 * (2 if two else 1)) / 1e6
        print('%s, %s: median %6.1f us  min %6.1f us   (%.0f MB: %.2f TB/s at the median)' % (
            'activation + gated copy' if two else 'one output', 'cold caches' if cold else 'back to back', t[0], t[1], mb, mb / t[0] / 1e6 * 1e6 / 1e6), flush=True)
