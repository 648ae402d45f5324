"""Per-shape timing of the implicit-GEMM conv kernel (and wgrad) on the shapes that dominate the
Pix2Pix GCC iteration at N=16, 256x256.  Usage: python scratch/bench_igemm.py [reps] [filter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import ops

DEV = 'cuda:0'
SHAPES = [
    # name, N, H, W, Ci, Co, k, s, p
    ('D.L2 128->256 k4s2 @128', 16, 128, 128, 128, 256, 4, 2, 1),
    ('D.L3 256->512 k4s2 @64', 16, 64, 64, 256, 512, 4, 2, 1),
    ('D.L4 512->1024 k4s1 @32', 16, 32, 32, 512, 1024, 4, 1, 1),
    ('D.L5 1024->1 k4s1 @31', 16, 31, 31, 1024, 1, 4, 1, 1),
    ('D.L1 6->128 k4s2 @256', 16, 256, 256, 6, 128, 4, 2, 1),
    ('tG.d1 64->128 k4s2 @128', 16, 128, 128, 64, 128, 4, 2, 1),
    ('tG.d2 128->256 k4s2 @64', 16, 64, 64, 128, 256, 4, 2, 1),
    ('tG.d3 256->512 k4s2 @32', 16, 32, 32, 256, 512, 4, 2, 1),
    ('tG.u3 adj 256->1024 k4s2 @32', 16, 32, 32, 256, 1024, 4, 2, 1),
    ('tG.u2 adj 128->512 k4s2 @64', 16, 64, 64, 128, 512, 4, 2, 1),
    ('tG.d4 512->512 k4s2 @16', 16, 16, 16, 512, 512, 4, 2, 1),
    ('tG.d5 512->512 k4s2 @8', 16, 8, 8, 512, 512, 4, 2, 1),
    ('tG.u4 adj 512->1024 k4s2 @16', 16, 16, 16, 512, 1024, 4, 2, 1),
    ('sG.d2 64->128 k4s2 @64', 16, 64, 64, 64, 128, 4, 2, 1),
    ('sG.d3 128->256 k4s2 @32', 16, 32, 32, 128, 256, 4, 2, 1),
    # thin first / last layers (<= 8 channels on one side): HBM-bound
    ('thin sG.d0 3->32 k4s2 @256', 16, 256, 256, 3, 32, 4, 2, 1),
    ('thin tG.d0 3->64 k4s2 @256', 16, 256, 256, 3, 64, 4, 2, 1),
    ('thin sG.u0 adj 3->64 k4s2 @256', 16, 256, 256, 3, 64, 4, 2, 1),
    ('thin tG.u0 adj 3->128 k4s2 @256', 16, 256, 256, 3, 128, 4, 2, 1),
    ('thin D.L1 6->128 k4s2 @256', 16, 256, 256, 6, 128, 4, 2, 1),
    ('thin D.L5 1024->1 k4s1 @31', 16, 31, 31, 1024, 1, 4, 1, 1),
    # K scan on the D.L2 geometry (M = 65536, N = 256): nk = 16 .. 128 k-steps per workgroup
    ('Kscan 64->256 k4s2 @128', 16, 128, 128, 64, 256, 4, 2, 1),
    ('Kscan 128->256 k4s2 @128', 16, 128, 128, 128, 256, 4, 2, 1),
    ('Kscan 256->256 k4s2 @128', 16, 128, 128, 256, 256, 4, 2, 1),
    ('Kscan 512->256 k4s2 @128', 16, 128, 128, 512, 256, 4, 2, 1),
    # SRGAN 96 -> 384 (N = 16): the discriminator's and VGG19's 3 x 3 layers  (python scratch/bench_igemm.py 10 SR)
    ('SR D 64->64 k3s2 @384', 16, 384, 384, 64, 64, 3, 2, 1),
    ('SR D 64->128 k3s1 @192', 16, 192, 192, 64, 128, 3, 1, 1),
    ('SR D 128->128 k3s2 @192', 16, 192, 192, 128, 128, 3, 2, 1),
    ('SR D 128->256 k3s1 @96', 16, 96, 96, 128, 256, 3, 1, 1),
    ('SR D 256->256 k3s2 @96', 16, 96, 96, 256, 256, 3, 2, 1),
    ('SR D 256->512 k3s1 @48', 16, 48, 48, 256, 512, 3, 1, 1),
    ('SR D 512->512 k3s2 @48', 16, 48, 48, 512, 512, 3, 2, 1),
    ('SR VGG 64->64 k3s1 @384', 16, 384, 384, 64, 64, 3, 1, 1),
    ('SR VGG 128->128 k3s1 @192', 16, 192, 192, 128, 128, 3, 1, 1),
    ('SR VGG 256->256 k3s1 @96', 16, 96, 96, 256, 256, 3, 1, 1),
    ('SR VGG 512->512 k3s1 @48', 16, 48, 48, 512, 512, 3, 1, 1),
    ('SR G 64->64 k3s1 @96', 16, 96, 96, 64, 64, 3, 1, 1),
    ('SR G 24->24 k3s1 @96', 16, 96, 96, 24, 24, 3, 1, 1),
    ('SR G 64->256 k3s1 @96', 16, 96, 96, 64, 256, 3, 1, 1),
    ('SR G 64->256 k3s1 @192', 16, 192, 192, 64, 256, 3, 1, 1),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    modes = (sys.argv[3] if len(sys.argv) > 3 else 'fdw')
    g = torch.Generator().manual_seed(0)
    print('%-32s %10s %10s %10s   (TFLOP/s; us)' % ('shape', 'fprop', 'dgrad', 'wgrad'))
    for name, N, H, W, Ci, Co, k, s, p in SHAPES:
        if filt and filt not in name:
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
        fl = 2.0 * N * Ho * Wo * Co * k * k * Ci
        out = []
        for tag, fn in (('f', lambda: ops.conv_fprop(x, w, Co, k, s, p, out=y)),
                        ('d', lambda: ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx)),
                        ('w', lambda: ops.conv_wgrad(x, dy, dw, k, s, p, accumulate=True))):
            if tag in modes:
                t = timeit(fn, reps)
                out.append('%5.0f/%4.0f' % (fl / t / 1e12, t * 1e6))
            else:
                out.append('    -     ')
        print('%-32s %10s %10s %10s' % (name, *out))


if __name__ == '__main__':
    main()
