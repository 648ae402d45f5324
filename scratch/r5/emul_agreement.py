"""CPU experiment: what update-direction agreement does bf16 storage ALONE give against the reference's
golden trajectory?  Runs the oracle with EMULATE_BF16 on the CycleGAN / SRGAN fixtures for the two golden iterations and feeds
its final weights into tests/_updates.MovementAgreement exactly as the GPU tests feed the HIP path's -- with and without
selecting on the measured gradient floor 3 * rms(g16 - g32) of iteration 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import gcc_oracle as O
from tests import _updates
from tests.golden.recipe import sample_idx
from tests.test_oracle_golden import build_cyclegan_oracle, _pre_norm_bias

z = np.load(os.path.join(ROOT, 'tests/golden/cyclegan_gcc.npz'))

def run(emulate, iters=2, lr0=False):
    O.EMULATE_BF16 = emulate
    try:
        m, t, opt = build_cyclegan_oracle(z)
        if lr0:
            for o in (m, t):
                o.lr_G = o.lr_D = o.lr_arch = 0.0
        init = {}
        for tag, who in (('s', m), ('t', t)):
            for w in 'AB':
                init[tag + 'G_' + w] = {k: v.detach().clone() for k, v in who.G[w].items()}
        grads = None
        for it in range(iters):
            m.set_input(torch.from_numpy(z['it%d.A' % it]), torch.from_numpy(z['it%d.B' % it]))
            m.optimize_parameters()
            if it == 0:
                grads = {}
                for tag, who in (('s', m), ('t', t)):
                    for w in 'AB':
                        for k in who.G_keys[w]:
                            grads[(tag + 'G_' + w, k)] = who.G[w][k].grad.clone()
            m.set_input(torch.from_numpy(z['it%d.vA' % it]), torch.from_numpy(z['it%d.vB' % it]))
            m.clipping_mask_alpha()
            m.optimizer_netD_arch()
        return m, t, init, grads
    finally:
        O.EMULATE_BF16 = False

_, _, _, g32 = run(False, 1, lr0=True)
_, _, _, g16 = run(True, 1, lr0=True)
m, t, init, _ = run(True, 2)
for use_floor in (False, True):
    agree = _updates.MovementAgreement()
    for tag, who in (('s', m), ('t', t)):
        for w in 'AB':
            net = tag + 'G_' + w
            for k in who.G_keys[w]:
                key = 'final.%s.%s' % (net, k)
                if key not in z.files or _pre_norm_bias(k) or 'running' in k:
                    continue
                ref = z[key].reshape(-1)
                idx = sample_idx(who.G[w][k].numel())
                got = who.G[w][k].detach().reshape(-1)[idx].numpy()
                ini = init[net][k].reshape(-1)[idx].numpy()
                mask = None
                if use_floor:
                    a, b = g32[(net, k)].reshape(-1), g16[(net, k)].reshape(-1)
                    floor = float((b - a).pow(2).mean().sqrt())
                    mask = (a.abs() >= 3 * floor)[idx].numpy()
                d_ref, d_got = ref - ini, got - ini
                sel = np.abs(d_ref) >= 0.5 * 2e-4 * 2
                if mask is not None:
                    sel &= mask
                a_ = agree.acc.setdefault(net[:2], [0, 0, 0, 0])
                a_[0] += int(sel.sum()); a_[1] += int((np.sign(d_got[sel]) == np.sign(d_ref[sel])).sum()); a_[2] += int((d_got[sel] != 0).sum()); a_[3] += d_ref.size
    print('floor selection' if use_floor else 'no floor selection')
    for tag, (n, ag, mv, tot) in sorted(agree.acc.items()):
        print('  %s: selected %d of %d, same direction %.4f' % (tag, n, tot, ag / max(n, 1)))
