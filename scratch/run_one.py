"""Run one conv op of one shape `reps` times (for rocprofv3 --pmc / --kernel-trace).  Usage: python3 scratch/run_one.py "<shape substring>" <f|d|w> [reps] [OPT=val ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import _lib, ops
from scratch.ab_shapes import SHAPES

DEV = 'cuda:0'
filt, tag = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for kv in sys.argv[4:]:
    k, v = kv.split('=')
    ops.lib().gcc_set_option(getattr(_lib, 'OPT_' + k), int(v))
g = torch.Generator().manual_seed(0)
trash = torch.empty(300 << 20, dtype=torch.uint8, device=DEV)
for name, N, H, W, Ci, Co, k, s, p in SHAPES:
    if filt not in name:
        continue
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = ops.new_act(N, Ci, H, W, DEV)
    x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, Ho, Wo, DEV)
    dy.copy_(torch.randn(N, Co, Ho, Wo, generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, Ho, Wo, DEV)
    dx = ops.new_act(N, Ci, H, W, DEV)
    dw = torch.zeros_like(m)
    fn = {'f': lambda: ops.conv_fprop(x, w, Co, k, s, p, out=y),
          'd': lambda: ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx),
          'w': lambda: ops.conv_wgrad(x, dy, dw, k, s, p, accumulate=True)}[tag]
    for r in range(reps):
        trash.fill_(r & 1)
        fn()
    torch.cuda.synchronize()
