"""VGG19 / SRResNet k3 s1 p1 layers at the SRGAN 96 -> 384 sizes (N = 16): tile plans of igemm_kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
DEV = 'cuda:0'
SHAPES = [('vgg 512->512 @48', 16, 48, 48, 512, 512), ('vgg 256->256 @96', 16, 96, 96, 256, 256), ('vgg 128->256 @192', 16, 192, 192, 128, 256),
          ('vgg 128->128 @192', 16, 192, 192, 128, 128), ('vgg 64->128 @192', 16, 192, 192, 64, 128), ('vgg 64->64 @384', 16, 384, 384, 64, 64),
          ('vgg 512->512 @24', 16, 24, 24, 512, 512), ('res 64->64 @96', 16, 96, 96, 64, 64), ('res 24->24 @96', 16, 96, 96, 24, 24)]
PLANS = [('default', {}), ('t128', dict(tile_families=1)), ('t256x128', dict(tile_families=2, big_min=1, big_nk=1)), ('t256x256', dict(tile_families=3, big_min=1, big_nk=1))]
def rate(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
print('%-20s' % 'shape' + ''.join('%26s' % p[0] for p in PLANS) + '   (fprop us [tile] / dgrad us [tile])')
for name, N, H, W, Ci, Co in SHAPES:
    x = ops.new_act(N, Ci, H, W, DEV); x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, H, W, DEV); dy.copy_(torch.randn(N, Co, H, W, generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, H, W, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
    d = _lib.conv_t(N, H, W, Ci, Co, 3, 3, 1, 1, (Ci + 7) & ~7, 0, (Co + 7) & ~7, 0)
    fl = 2.0 * N * H * W * Co * 9 * Ci
    row = '%-20s' % name
    for pname, plan in PLANS:
        ops.set_plan(**plan)
        d = ops.conv_desc(N, H, W, Ci, Co, 3, 1, 1, (Ci + 7) & ~7, (Co + 7) & ~7)
        tf, td = lib.gcc_conv_tile(C.byref(d), 0), lib.gcc_conv_tile(C.byref(d), 1)
        a = rate(lambda: ops.conv_fprop(x, w, Co, 3, 1, 1, out=y))
        b = rate(lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 3, 1, 1, out=dx))
        row += '  %6.0f[%6d]/%6.0f[%6d]' % (a, tf, b, td)
    ops.set_plan()
    print(row + '   ideal %5.0f us @2.5PF' % (fl / 2.5e15 * 1e6), flush=True)
