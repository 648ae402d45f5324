"""Checks that an optimizer step actually moved the weights the way the reference / the oracle says.

A bound such as |w_hip - w_ref| <= 2.2 * lr * steps is satisfied by a path that never updates (Adam moves a weight by
at most ~lr per step), so the model tests use these two instead:

* MovementAgreement (golden trajectories): over the sampled elements the REFERENCE moved by a clear amount
  (|w_ref_final - w_init| >= 0.5 * lr * steps, i.e. a gradient sign that did not flip between the steps), the HIP path
  must have moved too (>= 99 % non-zero displacement) and in the same direction (>= min_agree; the remainder are elements
  whose fp32 gradient is smaller than the bf16 pipeline's noise -- Adam turns even those into a full +-lr step).
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


class MovementAgreement:
    def __init__(self, min_move=0.5):
        self.min_move = min_move
        self.acc = {}

    def add(self, tag, init, got, ref, unit):
        """init / got / ref: sampled values (numpy); unit: lr * number of Adam updates of this tensor"""
        d_ref, d_hip = ref.reshape(-1) - init.reshape(-1), got.reshape(-1) - init.reshape(-1)
        sel = np.abs(d_ref) >= self.min_move * unit
        a = self.acc.setdefault(tag, [0, 0, 0, 0])
        a[0] += int(sel.sum())
        a[1] += int((np.sign(d_hip[sel]) == np.sign(d_ref[sel])).sum())
        a[2] += int((d_hip[sel] != 0).sum())
        a[3] += int(d_ref.size)

    def check(self, min_agree=0.9, min_moved=0.99, min_selected=0.1):
        assert self.acc, 'no tensors were compared'
        for tag, (n, agree, moved, total) in sorted(self.acc.items()):
            _report('update agreement %-9s: %6d of %6d sampled elements moved clearly in the reference; same direction %.4f, '
                    'moved at all %.4f' % (tag, n, total, agree / max(n, 1), moved / max(n, 1)))
        for tag, (n, agree, moved, total) in self.acc.items():
            assert n >= max(8, min_selected * total), (tag, 'too few clearly moved elements', n, total)
            assert moved >= min_moved * n, (tag, 'weights did not move', moved, n)
            assert agree >= min_agree * n, (tag, 'update direction disagrees with the reference', agree, n)


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
