"""Tile / split-K sweep of the implicit-GEMM conv on every U-Net layer shape of the headline config (student ngf 32, teacher
ngf 64, N = 16, 256 x 256): fprop and dgrad with BatchNorm partial statistics, 128-pixel tiles of 128 / 64 / 32 channels, K
splits 1 / 2 / 4 / 8 against the automatic plan.  Prints microseconds per call (the whole C call: split-K epilogue and channel
statistics kernels included) and the best configuration per shape.   python scratch/sweep_unet_tiles.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gcc_amd import _lib, ops

DEV = 'cuda:0'
L = ops.lib()


def shapes(ngf, tag):
    w = [ngf, ngf * 2, ngf * 4, ngf * 8, ngf * 8, ngf * 8, ngf * 8, ngf * 8]
    out = []
    for d in range(1, 8):            # down conv d: w[d-1] -> w[d] at input size 256 >> d
        out.append(('%s.d%d' % (tag, d), 16, 256 >> d, w[d - 1], w[d]))
    for d in range(1, 8):            # up conv d as its adjoint conv: big image 256 >> d with w[d-1] channels, small image with 2 w[d] (w[7] innermost)
        out.append(('%s.u%d' % (tag, d), 16, 256 >> d, w[d - 1], w[d] * (1 if d == 7 else 2)))
    return out


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
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, N, H, Ci, Co in shapes(32, 'sG') + shapes(64, 'tG'):
        x = ops.new_act(N, Ci, H, H, DEV); x.normal_()
        dy = ops.new_act(N, Co, H // 2, H // 2, DEV); dy.normal_()
        m = (torch.randn(Co, Ci, 4, 4, device=DEV) * 0.02).contiguous(memory_format=torch.channels_last)
        wp, wtp = ops.pack_weights(m)
        y = ops.new_act(N, Co, H // 2, H // 2, DEV)
        dx = ops.new_act(N, Ci, H, H, DEV)
        for mode in ('fprop', 'dgrad'):
            if mode == 'fprop':
                fn = lambda: ops.conv_fprop(x, wp, Co, 4, 2, 1, out=y, want_stats=True)
                flop = 2.0 * N * (H // 2) ** 2 * Co * 16 * Ci
            else:
                fn = lambda: ops.conv_dgrad(dy, wtp, Ci, H, H, 4, 2, 1, out=dx, want_stats=True)
                flop = 2.0 * N * (H // 2) ** 2 * Co * 16 * Ci
            res = {}
            for bc in (0, 128, 64, 32):
                cout = Co if mode == 'fprop' else Ci
                if bc and bc > max(16, 2 * cout):
                    continue
                for ks in (0, 1, 2, 4, 8):
                    L.gcc_set_option(_lib.OPT_IGEMM_FORCE_BC, bc)
                    L.gcc_set_option(_lib.OPT_IGEMM_FORCE_KSPLIT, ks)
                    try:
                        res[(bc, ks)] = timeit(fn, reps)
                    except Exception as e:      # noqa: BLE001
                        res[(bc, ks)] = float('nan')
            L.gcc_set_option(_lib.OPT_IGEMM_FORCE_BC, -1)
            L.gcc_set_option(_lib.OPT_IGEMM_FORCE_KSPLIT, -1)
            auto = res[(0, 0)]
            best = min((v, k) for k, v in res.items() if v == v)
            row = ' '.join('%d/%d:%.0f' % (k[0], k[1], v) for k, v in sorted(res.items()))
            print('%-7s %-5s H%3d %4d->%4d  auto %6.1f us (%6.1f TF/s)  best %6.1f us at BC %3d ks %d  | %s' % (
                name, mode, H, Ci, Co, auto, flop / auto / 1e6, best[0], best[1][0], best[1][1], row), flush=True)


if __name__ == '__main__':
    main()
