"""How far the bf16-EMULATING oracle's logged losses sit from the reference's fixtures, per family / iteration / scalar (CPU).
The numbers behind tests/test_oracle_golden.py::test_emulating_oracle_losses_stay_in_a_band_of_the_reference."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gcc_oracle as O
from tests import test_oracle_golden as T
from tests import _updates

G = os.path.join(ROOT, 'tests', 'golden')


def run(family):
    t0 = time.time()
    out = []
    for emulate in (False, True):
        O.EMULATE_BF16 = emulate
        try:
            if family == 'pix2pix':
                z = np.load(os.path.join(G, 'pix2pix_gcc_d6.npz'), allow_pickle=True); m, t, _ = T.build_gcc_oracle(z); ins = ('A', 'B', 'vA', 'vB')
            elif family == 'cyclegan':
                z = np.load(os.path.join(G, 'cyclegan_gcc.npz'), allow_pickle=True); m, t, _ = T.build_cyclegan_oracle(z); ins = ('A', 'B', 'vA', 'vB')
            elif family == 'sagan':
                z = np.load(os.path.join(G, 'sagan_gcc.npz'), allow_pickle=True); m, t, _ = T.build_sagan_oracle(z); ins = ('z', 'real', 'vz', 'vreal')
            else:
                z = np.load(os.path.join(G, 'srgan_gcc.npz'), allow_pickle=True); m, t, _ = T.build_srgan_oracle(z); ins = ('lr', 'hr', 'vlr', 'vhr')
            rows = []
            for it in range(2):
                m.set_input(torch.from_numpy(z['it%d.%s' % (it, ins[0])]), torch.from_numpy(z['it%d.%s' % (it, ins[1])]))
                m.optimize_parameters()
                m.set_input(torch.from_numpy(z['it%d.%s' % (it, ins[2])]), torch.from_numpy(z['it%d.%s' % (it, ins[3])]))
                m.clipping_mask_alpha()
                m.optimizer_netD_arch()
                for k in z.files:
                    for pre, who in (('it%d.loss.' % it, m), ('it%d.tloss.' % it, t)):
                        if k.startswith(pre):
                            rows.append((it, pre[-6], k[len(pre):], float(who.losses[k[len(pre):]]), float(z[k])))
            out.append(rows)
        finally:
            O.EMULATE_BF16 = False
    print('== %s (%.1f s)' % (family, time.time() - t0))
    for (it, w, name, v32, ref), (_, _, _, v16, _) in zip(*out):
        print('it%d %s %-24s ref %+.6g  fp32 oracle %+.6g  emulated %+.6g   |emul-ref| %.3g  rel %.3g' % (
            it, w, name, ref, v32, v16, abs(v16 - ref), abs(v16 - ref) / max(abs(ref), 1e-12)))


for f in sys.argv[1:] or ('pix2pix', 'cyclegan', 'sagan', 'srgan'):
    run(f)
