"""How fast does a chain of 100 tiny dependent kernels (one workgroup each) advance on stream B while stream A runs a chain of
igemm launches of a given grid?  (244 / 256 workgroups: every CU holds an igemm workgroup -- 2 waves x 256 VGPRs per SIMD, the
whole register file; 128 workgroups: half of the CUs are free.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcc_amd import ops

dev = torch.device('cuda:0')
ops.lib()
N = 16


def act(Cc, H, W):
    t = ops.new_act(N, Cc, H, W, dev)
    t.copy_(torch.randn(N, Cc, H, W, device=dev))
    return t


x4, y4 = act(512, 32, 32), ops.new_act(N, 1024, 31, 31, dev)
w4 = (torch.randn(1024, 16, 512, device=dev) * 0.02).to(torch.bfloat16)
dy4, dx4 = act(1024, 31, 31), ops.new_act(N, 512, 32, 32, dev)
wt4 = (torch.randn(512, 16, 1024, device=dev) * 0.02).to(torch.bfloat16)
x2, y2 = act(128, 128, 128), ops.new_act(N, 256, 64, 64, dev)
w2 = (torch.randn(256, 16, 128, device=dev) * 0.02).to(torch.bfloat16)
smalls = {1: torch.zeros(64, device=dev), 16: torch.zeros(16 * 256, device=dev), 128: torch.zeros(128 * 256, device=dev)}
from gcc_amd import _lib
bigs = {
    'L4 forward, 244 workgroups': (lambda: ops.conv_fprop(x4, w4, 1024, 4, 1, 1, out=y4), 12),
    'L2 forward, 256 workgroups': (lambda: ops.conv_fprop(x2, w2, 256, 4, 2, 1, out=y2), 40),
    'L4 data gradient, 128 workgroups': (lambda: ops.conv_dgrad(dy4, wt4, 512, 32, 32, 4, 1, 1, out=dx4), 8),
}
bigs['L2 forward as 256x128 tiles, 512 workgroups'] = (lambda: ops.conv_fprop(x2, w2, 256, 4, 2, 1, out=y2), 30)
bigs['L2 forward as 128x128 tiles, 1024 workgroups'] = (lambda: ops.conv_fprop(x2, w2, 256, 4, 2, 1, out=y2), 30)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
ev = lambda: torch.cuda.Event(enable_timing=True)
for name, (big, reps) in bigs.items():
  ops.set_plan(tile_families=2 if '256x128' in name else (1 if '128x128' in name else 3))
  for nwg, small in smalls.items():
    for _ in range(2):
          torch.cuda.synchronize()
          a0, a1, b0, b1 = ev(), ev(), ev(), ev()
          rel = torch.cuda.Event()
          with torch.cuda.stream(sA):
              a0.record()
              big()
              rel.record()
              for _ in range(reps):
                  big()
              a1.record()
          with torch.cuda.stream(sB):
              sB.wait_event(rel)
              b0.record()
              for _ in range(100):
                  ops.fill(small, 1.0)
              b1.record()
          torch.cuda.synchronize()
    print('%-46s A: %2d launches in %.2f ms | B: 100 kernels of %3d workgroups in %.2f ms (%.1f us each)' % (
        name, reps + 1, a0.elapsed_time(a1), nwg, b0.elapsed_time(b1), b0.elapsed_time(b1) * 10))
for nwg, small in smalls.items():
    torch.cuda.synchronize()
    b0, b1 = ev(), ev()
    with torch.cuda.stream(sB):
        b0.record()
        for _ in range(100):
            ops.fill(small, 1.0)
        b1.record()
    torch.cuda.synchronize()
    print('B alone: 100 kernels of %3d workgroups in %.2f ms' % (nwg, b0.elapsed_time(b1)))
