"""single-launch durations of the U-Net outer layers (student ngf 32, teacher ngf 64, N = 16) under tile / stage / no-load ablations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_amd import ops, _lib
lib = _lib.load()
DEV = 'cuda:0'
SHAPES = [  # name, N, H, W, Ci, Co (conv geometry k4 s2 p1; 'd' = its data gradient = the up layer's transposed conv)
    ('sG.d1 32->64 @128', 16, 128, 128, 32, 64, 'f'),
    ('sG.d2 64->128 @64', 16, 64, 64, 64, 128, 'f'),
    ('sG.d3 128->256 @32', 16, 32, 32, 128, 256, 'f'),
    ('sG.u1 adj 32<-128 @128', 16, 128, 128, 32, 128, 'd'),
    ('sG.u2 adj 64<-256 @64', 16, 64, 64, 64, 256, 'd'),
    ('sG.u3 adj 128<-512 @32', 16, 32, 32, 128, 512, 'd'),
    ('tG.d1 64->128 @128', 16, 128, 128, 64, 128, 'f'),
    ('tG.u1 adj 64<-256 @128', 16, 128, 128, 64, 256, 'd'),
    ('tG.u2 adj 128<-512 @64', 16, 64, 64, 128, 512, 'd'),
]
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
CFGS = [('default', {}), ('stages2', {_lib.OPT_IGEMM_STAGES: 2}), ('noloads', {'diag': 2}),      # diagnostic build only (GCC_HIP_LIB=gcc_amd/libgcc_hip_diag.so): skipped on the shipped library
        
        ('bc32', {_lib.OPT_IGEMM_FORCE_BC: 32}), ('bc64', {_lib.OPT_IGEMM_FORCE_BC: 64}), ('bc128', {_lib.OPT_IGEMM_FORCE_BC: 128}),
        ('nohalo', {_lib.OPT_IGEMM_HALO: 0})]
print('%-26s' % 'shape' + ''.join('%16s' % c[0] for c in CFGS) + '   roofline us (bytes @ 4 TB/s | flop @ 2.5 PF)')
for name, N, H, W, Ci, Co, mode in SHAPES:
    Ho, Wo = H // 2, W // 2
    x = ops.new_act(N, Ci, H, W, DEV); x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, Ho, Wo, DEV); dy.copy_(torch.randn(N, Co, Ho, Wo, generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co, Ci, 4, 4, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    y = ops.new_act(N, Co, Ho, Wo, DEV); dx = ops.new_act(N, Ci, H, W, DEV)
    fn = (lambda: ops.conv_fprop(x, w, Co, 4, 2, 1, out=y)) if mode == 'f' else (lambda: ops.conv_dgrad(dy, wt, Ci, H, W, 4, 2, 1, out=dx))
    row = '%-26s' % name
    for cname, opts in CFGS:
        if 'diag' in opts and not hasattr(lib, 'gcc_diag_set'):
            row += '%16s' % 'n/a'
            continue
        for k, v in opts.items(): (lib.gcc_diag_set(v) if k == 'diag' else lib.gcc_set_option(k, v))
        try:
            a, b = med(fn)
            row += '%9.1f/%6.1f' % (a, b)
        except Exception as e:
            row += '%16s' % 'err'
        for k in opts: (lib.gcc_diag_set(0) if k == 'diag' else lib.gcc_set_option(k, -1))
    nbytes = 2.0 * (N * H * W * Ci + N * Ho * Wo * Co + 16 * Ci * Co)
    fl = 2.0 * N * Ho * Wo * Co * 16 * Ci
    print(row + '   %5.1f | %5.1f' % (nbytes / 4e12 * 1e6, fl / 2.5e15 * 1e6), flush=True)
