"""Shader clock inside igemm_kernel's main loop (diagnostic library built by scratch/probes/build_clock_probe.sh):
d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup, after >= 2 s of back-to-back launches on random data
(MI355X_MICROARCH.md, 'DVFS give-back' item 6).  Full-chip against half-chip launches of the PatchGAN's L4."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'probes', sys.argv[1] if len(sys.argv) > 1 else 'libgcc_hip_probe_rot1.so')
print('library', os.path.basename(_lib.LIB_PATH))
from gcc_amd import ops

lib = ops.lib()
lib.gcc_probe_read.restype = C.c_int
lib.gcc_probe_read.argtypes = [C.c_void_p, C.c_int]
dev = torch.device('cuda:0')
torch.manual_seed(0)


def act(N, Cc, H, W):
    t = ops.new_act(N, Cc, H, W, dev)
    t.copy_(torch.randn(N, Cc, H, W, device=dev))
    return t


def run(name, fn, seconds=2.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    lib.gcc_probe_read(buf.ctypes.data, 1)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    lib.gcc_probe_read(buf.ctypes.data, 0)
    m = buf[buf[:, 3] == 1]
    clk = m[:, 0].astype(np.float64) / m[:, 1].astype(np.float64) * 100.0        # MHz
    kstep = m[:, 1].astype(np.float64) / 100.0 / m[:, 2].astype(np.float64)     # us per k-step
    print('%-44s %4d workgroups  nk %3d  launch %7.1f us | clock median %4.0f MHz (min %4.0f max %4.0f) | '
          'k-step median %.3f us (min %.3f max %.3f) | loop %.1f us' % (
              name, len(m), int(m[0, 2]), us, np.median(clk), clk.min(), clk.max(), np.median(kstep), kstep.min(), kstep.max(),
              float(np.median(m[:, 1])) / 100.0), flush=True)


N = 16
# PatchGAN L4: 512 -> 1024, k4 s1 p1, 32x32 -> 31x31
x4 = act(N, 512, 32, 32)
w4 = torch.randn(1024, 16, 512, device=dev).mul_(0.02).to(torch.bfloat16)
y4 = ops.new_act(N, 1024, 31, 31, dev)
dy4 = act(N, 1024, 31, 31)
wt4 = torch.randn(512, 16, 1024, device=dev).mul_(0.02).to(torch.bfloat16)
dx4 = ops.new_act(N, 512, 32, 32, dev)
# L3: 256 -> 512, k4 s2 p1, 64x64 -> 32x32 ; L2: 128 -> 256, 128x128 -> 64x64
x3 = act(N, 256, 64, 64)
w3 = torch.randn(512, 16, 256, device=dev).mul_(0.02).to(torch.bfloat16)
y3 = ops.new_act(N, 512, 32, 32, dev)
x2 = act(N, 128, 128, 128)
w2 = torch.randn(256, 16, 128, device=dev).mul_(0.02).to(torch.bfloat16)
y2 = ops.new_act(N, 256, 64, 64, dev)

run('L4 fprop (244 workgroups of 256x256)', lambda: ops.conv_fprop(x4, w4, 1024, 4, 1, 1, out=y4))
run('L4 dgrad (128 workgroups of 256x256)', lambda: ops.conv_dgrad(dy4, wt4, 512, 32, 32, 4, 1, 1, out=dx4))
ops.set_plan(pair=1)
run('L4 dgrad, pair split (256 workgroups, K/2)', lambda: ops.conv_dgrad(dy4, wt4, 512, 32, 32, 4, 1, 1, out=dx4))
ops.set_plan(pair=0)
run('L3 fprop (128 workgroups of 256x256)', lambda: ops.conv_fprop(x3, w3, 512, 4, 2, 1, out=y3))
run('L2 fprop (256 workgroups of 256x256)', lambda: ops.conv_fprop(x2, w2, 256, 4, 2, 1, out=y2))
# zero-filled operands: the clock the chip holds when the MFMAs toggle nothing
x4.zero_()
run('L4 fprop, zero-filled pixels', lambda: ops.conv_fprop(x4, w4, 1024, 4, 1, 1, out=y4))
