"""Where a launch of igemm_kernel spends its time (diagnostic library of scratch/probes/build_clock_probe.sh): per workgroup the
100 MHz realtime stamps at kernel entry, k loop start, k loop end and kernel end (after the epilogue's stores are acknowledged)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'scratch', 'probes', 'libgcc_hip_probe_rot0.so')
from gcc_amd import ops
lib = ops.lib()
lib.gcc_probe_read.restype = C.c_int
lib.gcc_probe_read.argtypes = [C.c_void_p, C.c_int]
DEV = 'cuda:0'
SHAPES = [
    ('sG.d1 32->64 @128', 16, 128, 128, 32, 64, 'f'),
    ('sG.d2 64->128 @64', 16, 64, 64, 64, 128, 'f'),
    ('sG.d3 128->256 @32', 16, 32, 32, 128, 256, 'f'),
    ('sG.u1 adj 32<-128 @128', 16, 128, 128, 32, 128, 'd'),
    ('sG.u2 adj 64<-256 @64', 16, 64, 64, 64, 256, 'd'),
    ('tG.d1 64->128 @128', 16, 128, 128, 64, 128, 'f'),
    ('tG.u1 adj 64<-256 @128', 16, 128, 128, 64, 256, 'd'),
]
g = torch.Generator().manual_seed(0)
lib.gcc_set_option(_lib.OPT_IGEMM_HALO, 0)        # the stamps live in igemm_kernel
for stages in (3, 2):
    lib.gcc_set_option(_lib.OPT_IGEMM_STAGES, stages)
    print('== GCC_IGEMM_STAGES = %d' % stages)
    for name, N, H, W, Ci, Co, mode in SHAPES:
        Ho, Wo = H // 2, W // 2
        x = ops.new_act(N, Ci, H, W, DEV); x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
        dy = ops.new_act(N, Co, Ho, Wo, DEV); dy.copy_(torch.randn(N, Co, Ho, Wo, generator=g).bfloat16().to(DEV))
        m = (torch.randn(Co, Ci, 4, 4, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
        w, wt = ops.pack_weights(m)
        y = ops.new_act(N, Co, Ho, Wo, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
        fn = (lambda: ops.conv_fprop(x, w, Co, 4, 2, 1, out=y)) if mode == 'f' else (lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 4, 2, 1, out=dx))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), dtype=np.uint64)
        lib.gcc_probe_read(buf.ctypes.data, 1)
        fn(); torch.cuda.synchronize()
        lib.gcc_probe_read(buf.ctypes.data, 0)
        mm = buf[buf[:, 3] == 1].astype(np.float64)
        if not len(mm):
            print('%-24s no stamps' % name); continue
        t0 = mm[:, 4].min()
        ent, l0, l1, end = [(mm[:, c] - t0) / 100.0 for c in (4, 5, 6, 7)]      # us
        print('%-24s %4d wgs nk %3d | span %5.1f us | entry spread p50 %4.1f max %4.1f | entry->loop %4.1f | loop %5.1f (%.2f/step) | '
              'loop end->kernel end %4.1f | wg life p50 %5.1f max %5.1f' % (
                  name, len(mm), int(mm[0, 2]), end.max(), np.median(ent), ent.max(), np.median(l0 - ent), np.median(l1 - l0),
                  np.median(l1 - l0) / mm[0, 2], np.median(end - l1), np.median(end - ent), (end - ent).max()), flush=True)
