"""Halo kernel (conv_halo.hip) against igemm_kernel and an fp32 torch reference on the PatchGAN shapes, statistics included."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import _lib, ops

lib = ops.lib()
DEV = 'cuda:0'
g = torch.Generator().manual_seed(1)


def run(N, H, W, Ci, Co, s):
    k, p = 4, 1
    Ho, Wo = (H + 2 - k) // s + 1, (W + 2 - k) // s + 1
    x = ops.new_act(N, Ci, H, W, DEV)
    x.copy_(torch.randn(N, Ci, H, W, generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, Ho, Wo, DEV)
    dy.copy_(torch.randn(N, Co, Ho, Wo, generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    outs = {}
    for halo in (0, 2):
        lib.gcc_set_option(_lib.OPT_IGEMM_HALO, halo)
        y = ops.new_act(N, Co, Ho, Wo, DEV)
        y.fill_(7.0)
        dx = ops.new_act(N, Ci, H, W, DEV)
        dx.fill_(7.0)
        _, st = ops.conv_fprop(x, w, Co, k, s, p, out=y, want_stats=True)
        ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx)
        torch.cuda.synchronize()
        outs[halo] = (y.float().clone(), dx.float().clone(), st.double().sum(0).clone(), st.shape[0])
    ref_y = torch.nn.functional.conv2d(x.float(), m.bfloat16().float(), stride=s, padding=1)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), m.bfloat16().float(), dy.float(), stride=s, padding=1)
    for name, i, ref in (('fprop', 0, ref_y), ('dgrad', 1, ref_dx)):
        a, b = outs[0][i], outs[2][i]
        sc = ref.abs().max().item()
        print(f'{N}x{Ci}->{Co}@{H} s{s} {name}: igemm-vs-ref {((a - ref).abs().max() / sc).item():.2e} halo-vs-ref '
              f'{((b - ref).abs().max() / sc).item():.2e} halo-vs-igemm {((a - b).abs().max() / sc).item():.2e}  differing '
              f'{(a != b).float().mean().item():.4f}')
    yb = outs[2][0].double()
    want = torch.stack([yb.sum((0, 2, 3)), (yb * yb).sum((0, 2, 3))])
    got = outs[2][2]
    print(f'   statistics rows {outs[0][3]} / {outs[2][3]}: halo sums vs sums of its own output, rel '
          f'{((got - want).abs().max() / want.abs().max()).item():.2e}')


run(16, 128, 128, 128, 256, 2)
run(16, 64, 64, 256, 512, 2)
run(16, 32, 32, 512, 1024, 1)
run(3, 32, 32, 64, 256, 1)
run(16, 16, 16, 64, 256, 1)
