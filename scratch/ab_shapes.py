"""Interleaved A/B timing of conv launches under option variants, one process (cdna_hip_programming.md rule 24).
Usage: python scratch/ab_shapes.py <modes fdw> <filter> <variant>...     variant = name:OPT=val,OPT=val  (OPT = _lib.OPT_* suffix)
e.g.   python scratch/ab_shapes.py w "D.L" base:WGRAD_XCD_SPLIT=0 xcd:WGRAD_XCD_SPLIT=1
Between two timed launches of a shape a 300 MB buffer is rewritten so that the launch starts with its operands beyond the
L2s (as in the training step, where other layers ran in between); pass cold=0 as a variant-less token to skip that."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import _lib, ops

DEV = 'cuda:0'
SHAPES = [
    ('D.L2 128->256 k4s2 @128', 16, 128, 128, 128, 256, 4, 2, 1),
    ('D.L3 256->512 k4s2 @64', 16, 64, 64, 256, 512, 4, 2, 1),
    ('D.L4 512->1024 k4s1 @32', 16, 32, 32, 512, 1024, 4, 1, 1),
    ('tG.d1 64->128 k4s2 @128', 16, 128, 128, 64, 128, 4, 2, 1),
    ('tG.d2 128->256 k4s2 @64', 16, 64, 64, 128, 256, 4, 2, 1),
    ('tG.d3 256->512 k4s2 @32', 16, 32, 32, 256, 512, 4, 2, 1),
    ('tG.u3 adj 256->1024 k4s2 @32', 16, 32, 32, 256, 1024, 4, 2, 1),
    ('tG.u2 adj 128->512 k4s2 @64', 16, 64, 64, 128, 512, 4, 2, 1),
    ('tG.d4 512->512 k4s2 @16', 16, 16, 16, 512, 512, 4, 2, 1),
    ('tG.u4 adj 512->1024 k4s2 @16', 16, 16, 16, 512, 1024, 4, 2, 1),
    ('sG.d2 64->128 k4s2 @64', 16, 64, 64, 64, 128, 4, 2, 1),
    ('sG.d3 128->256 k4s2 @32', 16, 32, 32, 128, 256, 4, 2, 1),
    ('sG.u3 adj 128->512 k4s2 @32', 16, 32, 32, 128, 512, 4, 2, 1),
]


def main():
    modes = sys.argv[1] if len(sys.argv) > 1 else 'fdw'
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    cold = True
    variants = []
    for tok in sys.argv[3:]:
        if tok == 'cold=0':
            cold = False
            continue
        name, _, rest = tok.partition(':')
        opts = []
        for kv in filter(None, rest.split(',')):
            k, v = kv.split('=')
            opts.append((getattr(_lib, 'OPT_' + k), int(v)))
        variants.append((name, opts))
    if not variants:
        variants = [('default', [])]
    lib = ops.lib()
    rounds = 12
    g = torch.Generator().manual_seed(0)
    trash = torch.empty(300 << 20, dtype=torch.uint8, device=DEV)
    print('%-30s %-5s ' % ('shape', 'op') + ' '.join('%-22s' % v[0] for v in variants) + '  (median us / min us / TFLOP/s at median)')
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
        fns = {'f': lambda: ops.conv_fprop(x, w, Co, k, s, p, out=y),
               'd': lambda: ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx),
               'w': lambda: ops.conv_wgrad(x, dy, dw, k, s, p, accumulate=True)}
        for tag in modes:
            fn = fns[tag]
            times = {v[0]: [] for v in variants}
            for r in range(rounds + 2):
                for vname, opts in variants:
                    prev = [(o, lib.gcc_set_option(o, val)) for o, val in opts]
                    if cold:
                        trash.fill_(r & 1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    for o, pv in prev:
                        lib.gcc_set_option(o, pv)
                    if r >= 2:
                        times[vname].append(e0.elapsed_time(e1) * 1e3)
            cells = []
            for vname, _ in variants:
                t = sorted(times[vname])
                med = t[len(t) // 2]
                cells.append('%7.1f /%7.1f /%5.0f' % (med, t[0], fl / med / 1e6))
            print('%-30s %-5s ' % (name, {'f': 'fprop', 'd': 'dgrad', 'w': 'wgrad'}[tag]) + ' '.join('%-22s' % c for c in cells), flush=True)


if __name__ == '__main__':
    main()
