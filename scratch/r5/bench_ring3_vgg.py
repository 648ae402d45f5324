"""VGG conv1_2 shape through the ring-walk kernel: the three call forms in several orders (is a difference the form's or the order's?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops, _lib
DEV = torch.device('cuda:0')
def med(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]
g = torch.Generator().manual_seed(0)
N, H, W, Ci, Co = 16, 384, 384, 64, 64
x = ops.new_act(N, Ci, H, W, DEV); x.normal_()
dy = ops.new_act(N, Co, H, W, DEV); dy.normal_()
m = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
b = torch.zeros(Co, device=DEV)
w, wt = ops.pack_weights(m)
y = ops.new_act(N, Co, H, W, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
forms = {
    'stats': lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, want_stats=True),
    'plain': lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y),
    'bias+relu': lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, bias=b, act=ops.ACT_RELU),
    'bias': lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, bias=b),
    'relu': lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y, act=ops.ACT_RELU),
    'dgrad': lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 3, 1, 1, out=dx),
    'dgrad(x as dy)': lambda: ops.conv_dgrad(x, wt, Ci, H, W, 3, 1, 1, out=dx),
}
for order in (('stats', 'plain', 'bias+relu', 'bias', 'relu', 'dgrad', 'dgrad(x as dy)'), ('dgrad', 'relu', 'bias+relu', 'plain', 'stats', 'bias', 'dgrad(x as dy)')):
    print('  '.join('%s %.1f/%.1f' % ((k,) + med(forms[k])) for k in order), flush=True)
