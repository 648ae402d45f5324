"""Checks that an optimizer step actually moved the weights the way the reference / the oracle says.

A bound such as |w_hip - w_ref| <= 2.2 * lr * steps is satisfied by a path that never updates (Adam moves a weight by
at most ~lr per step), so the model tests use these two instead:

* MovementAgreement (golden trajectories): over the sampled elements the REFERENCE moved by a clear amount
  (|w_ref_final - w_init| >= 0.5 * lr * steps, i.e. a gradient sign that did not flip between the steps) AND whose fp32
  oracle gradient of the first iteration stands clear of the bf16 pipeline's noise in its tensor (floor_masks: |g32| >=
  3 * rms(g_bf16_emulated - g32), the selection sign_check uses), the HIP path must have moved too (>= 99 % non-zero
  displacement) and in the same direction (>= 0.98; measured 0.991-0.9999, profiles/r5end_test_report.txt).  Without the floor selection the figure is reported, not judged: Adam
  turns an element whose gradient is below the noise into a full +-lr step of arbitrary sign, and the bf16-EMULATING ORACLE
  itself agrees with the reference on only 0.902-0.904 of the clearly moved generator weights of the CycleGAN fixture
  (scratch/r5/emul_agreement.py, CPU; the HIP path: 0.901-0.906) against 0.9993 above the floor.
* sign_check (oracle gradients): for elements whose fp32 oracle gradient exceeds the measured bf16 floor of its tensor
  (|g32| >= 3 * rms(g_bf16_emulated - g32)) the displacement of one real Adam step must be opposite to the gradient on
  >= 99 %.
"""
import numpy as np
import torch

from tests.golden.recipe import sample_idx


def _report(line):
    """stdout (shown by pytest on failure) and, when GCC_TEST_REPORT names a file, appended there: the measured agreement
    of passing runs is evidence too"""
    import os
    print(line)
    path = os.environ.get('GCC_TEST_REPORT')
    if path:
        with open(path, 'a') as fh:
            fh.write('%s | %s\n' % (os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0], line))


def sampled(t):
    g = t.detach().float().cpu().reshape(-1)
    return g[sample_idx(g.numel())].numpy().copy()


def snapshot(nets):
    """{tag: module} -> {tag: {name: sampled initial values}}"""
    return {tag: {k: sampled(v) for k, v in net.state_dict().items() if v.dtype.is_floating_point} for tag, net in nets.items()}


def floor_masks(g32, g16, k=3.0):
    """{key: fp32 oracle gradient}, {key: bf16-emulated oracle gradient} -> {key: bool mask over the sampled elements}: those
    whose gradient is at least k times the rms deviation bf16 storage alone causes in that tensor"""
    out = {}
    for key, a in g32.items():
        a, b = a.detach().float().cpu().reshape(-1), g16[key].detach().float().cpu().reshape(-1)
        floor = float((b - a).pow(2).mean().sqrt())
        m = a.abs() >= max(k * floor, 1e-30)
        out[key] = m[sample_idx(m.numel())].numpy()
    return out


class MovementAgreement:
    def __init__(self, min_move=0.5):
        self.min_move = min_move
        self.acc = {}
        self.raw = {}
        self.emu = {}         # the bf16-emulating oracle's own agreement with the reference over the same clearly moved elements

    def add(self, tag, init, got, ref, unit, mask=None, emul=None):
        """init / got / ref: sampled values (numpy); unit: lr * number of Adam updates of this tensor; mask: floor_masks()
        entry of the tensor (None: every clearly moved element is judged); emul: the same sampled elements after the same
        iterations on the bf16-emulating oracle (the bar on ALL clearly moved elements is tied to the figure the
        precision itself reaches, not to a constant)"""
        d_ref, d_hip = ref.reshape(-1) - init.reshape(-1), got.reshape(-1) - init.reshape(-1)
        clear = np.abs(d_ref) >= self.min_move * unit
        for acc, sel in ((self.raw, clear), (self.acc, clear if mask is None else (clear & mask.reshape(-1)))):
            a = acc.setdefault(tag, [0, 0, 0, 0])
            a[0] += int(sel.sum())
            a[1] += int((np.sign(d_hip[sel]) == np.sign(d_ref[sel])).sum())
            a[2] += int((d_hip[sel] != 0).sum())
            a[3] += int(d_ref.size)
        if emul is not None:
            d_emu = np.asarray(emul).reshape(-1) - init.reshape(-1)
            e = self.emu.setdefault(tag, [0, 0])
            e[0] += int(clear.sum())
            e[1] += int((np.sign(d_emu[clear]) == np.sign(d_ref[clear])).sum())

    def check(self, min_agree=0.98, min_moved=0.99, min_selected=0.04, emul_margin=0.02):
        assert self.acc, 'no tensors were compared'
        for tag, (n, agree, moved, total) in sorted(self.acc.items()):
            rn, ragree = self.raw[tag][0], self.raw[tag][1]
            en, eagree = self.emu.get(tag, (0, 0))
            _report('update agreement %-9s: %6d of %6d sampled elements moved clearly in the reference and stand above the bf16 '
                    'gradient floor; same direction %.4f, moved at all %.4f  (all %d clearly moved ones: %.4f%s)' % (
                        tag, n, total, agree / max(n, 1), moved / max(n, 1), rn, ragree / max(rn, 1),
                        '; the bf16-emulating oracle on the same elements: %.4f' % (eagree / max(en, 1)) if en else ''))
        for tag, (n, agree, moved, total) in self.acc.items():
            assert n >= max(8, min_selected * total), (tag, 'too few clearly moved elements above the floor', n, total)
            assert moved >= min_moved * n, (tag, 'weights did not move', moved, n)
            assert agree >= min_agree * n, (tag, 'update direction disagrees with the reference', agree, n)
            rn, ragree = self.raw[tag][0], self.raw[tag][1]
            en, eagree = self.emu.get(tag, (0, 0))
            # all clearly moved elements: no worse than the emulating oracle itself by more than emul_margin (where the test ran it
            # over this tensor class), and never below 0.85
            bar = max(0.85, eagree / en - emul_margin) if en >= 64 else 0.85
            assert ragree >= bar * rn, (tag, 'update direction disagrees with the reference (all clearly moved)', ragree / max(rn, 1), bar)


def sign_check(tag, before, after, g32, g16, acc):
    """one real Adam step: displacement against the oracle's gradient on elements above the bf16 floor of the tensor"""
    g32, g16 = g32.detach().float().cpu().reshape(-1), g16.detach().float().cpu().reshape(-1)
    d = (after.detach().float().cpu() - before.detach().float().cpu()).reshape(-1)
    floor = float((g16 - g32).pow(2).mean().sqrt())
    sel = g32.abs() >= max(3.0 * floor, 1e-3 * float(g32.abs().max()), 1e-30)
    a = acc.setdefault(tag, [0, 0, 0])
    a[0] += int(sel.sum())
    a[1] += int((torch.sign(d[sel]) == -torch.sign(g32[sel])).sum())
    a[2] += int(g32.numel())


def sign_report(acc, min_frac=0.99, min_selected=0.02):
    assert acc
    for tag, (n, ok, total) in sorted(acc.items()):
        _report('update sign %-6s: %8d of %8d elements above the bf16 floor; step opposite to the oracle gradient on %.4f' % (
            tag, n, total, ok / max(n, 1)))
    for tag, (n, ok, total) in acc.items():
        assert n >= min_selected * total, (tag, 'too few elements above the floor', n, total)
        assert ok >= min_frac * n, (tag, 'update sign disagrees with the oracle gradient', ok, n)


MAP_LOSSES = ('G_GAN', 'D_real', 'D_fake', 'D_arch', 'D_arch_diff', 'teacher_D_arch_diff', 'D_fake_arch', 'D_real_arch')


def loss_tol(name, ref, n_map, rel=3e-2):
    """Bar on |got - ref| of a logged loss scalar (no max(1, |ref|) denominator): `rel` of the reference's
    value, but not below what ONE flipped decision costs.  The GAN / arch terms are means over the n_map outputs of a PatchGAN
    map behind a hinge (or differences of two such means): an output that bf16 rounding carries across a kink, here or in a
    LeakyReLU / gate below it, moves the mean by up to ~2 / n_map (72-value map of the 64 x 64 fixtures: 0.028; the 900 values
    of a 256 x 256 image: 0.0022).  Every other term is a mean over >= 1e4 elements: floor 1e-3."""
    floor = 2.0 / max(int(n_map), 1) if any(name == m or name.startswith(m) for m in MAP_LOSSES) else 1e-3
    return max(rel * abs(ref), floor)


# ---- logged-loss bars, reported per oracle ------------------------------------------------------------------
# Every logged scalar of a model test is judged against TWO values: the reference's fixture (or the fp32 oracle where a test has no
# fixture at its size) and the bf16-EMULATING oracle (oracle.EMULATE_BF16: the reference's arithmetic with a rounding at every point
# the HIP path stores bf16).  The emulating mode is pinned on the CPU: tests/test_oracle_golden.py::
# test_emulating_oracle_losses_stay_in_a_band_of_the_reference holds it inside a stated band of the reference's own fixtures.
# LossBars keeps the two errors APART: a scalar that passes only against the emulating oracle is named in the report, and each
# family states how many such scalars it tolerates.
LOSS_BAR_LOG = []          # (family, label, e_ref, e_emul): read by conftest.pytest_sessionfinish for the session summary


class LossBars:
    def __init__(self, family, n_map, rel=3e-2):
        self.family, self.n_map, self.rel, self.rows = family, n_map, rel, []

    def add(self, label, name, got, ref, emul=None, n_map=None):
        """label: 'it0 S G_GAN'-style tag; errors are in units of loss_tol's bar (<= 1 passes)"""
        nm = self.n_map if n_map is None else n_map
        e_ref = abs(got - ref) / loss_tol(name, ref, nm, self.rel)
        e_emul = abs(got - emul) / loss_tol(name, emul, nm, self.rel) if emul is not None else float('inf')
        self.rows.append((label, got, ref, emul, e_ref, e_emul))
        LOSS_BAR_LOG.append((self.family, label, e_ref, e_emul))
        return e_ref, e_emul

    def check(self, max_emul_only=0.10, require=True):
        """require: every scalar within the bar of the reference OR of the emulating oracle (False: the caller keeps its own
        asserts and this only reports).  max_emul_only: the share of scalars that may pass through the emulating oracle alone."""
        n = len(self.rows)
        assert n, 'no loss scalars were compared'
        ref_ok = [r for r in self.rows if r[4] <= 1.0]
        emul_only = [r for r in self.rows if r[4] > 1.0 and r[5] <= 1.0]
        neither = [r for r in self.rows if r[4] > 1.0 and r[5] > 1.0]
        worst_ref = max(r[4] for r in ref_ok) if ref_ok else float('nan')
        _report('logged losses [%s]: %d scalars; %d within the bar of the reference (worst |err| / bar %.3f); %d only within the bar '
                'of the bf16-emulating oracle; %d within neither (bar: %.0e relative, floor 2 / %d on PatchGAN-map means, 1e-3 else)' % (
                    self.family, n, len(ref_ok), worst_ref, len(emul_only), len(neither), self.rel, self.n_map))
        for label, got, ref, emul, e_ref, e_emul in emul_only + neither:
            _report('   %s [%s] %s: got %.6g  reference %.6g (|err| / bar %.2f)  emulating oracle %s (|err| / bar %.2f)' % (
                'EMUL-ONLY' if e_emul <= 1.0 else 'NEITHER', self.family, label, got, ref, e_ref,
                '%.6g' % emul if emul is not None else '-', e_emul))
        if require:
            assert not neither, [(r[0], r[1], r[2], r[3]) for r in neither]
        assert len(emul_only) <= max_emul_only * n, ('too many scalars pass only against the emulating oracle',
                                                     [r[0] for r in emul_only], n)
