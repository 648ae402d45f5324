"""Deterministic weight recipe shared by tests/golden/make_fixtures.py (which loads these values
INTO the real reference models before running them) and by the tests (which rebuild the same
values instead of storing megabytes of random weights).  Pure torch-CPU; imports nothing from the
reference."""
import zlib

import torch


def recipe_state_dict(shapes, seed):
    """name -> tensor following the reference's init rule (utils/util.py:261-286) with one seeded
    generator per tensor: conv W ~ N(0,.02), conv bias 0, BN gamma ~ N(1,.02), BN beta ~ N(0,1),
    running_mean ~ N(0,.1), running_var ~ U(.5,1.5), alpha = 1, num_batches_tracked = 0."""
    sd = {}
    for k, shp in shapes.items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(k.encode()) % 1000003)
        shp = tuple(shp)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif k.endswith('running_var'):
            sd[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith('alpha'):
            sd[k] = torch.ones(shp)
        elif len(shp) == 4:
            sd[k] = torch.randn(shp, generator=g) * 0.02
        elif k.endswith('.weight'):
            sd[k] = 1.0 + torch.randn(shp, generator=g) * 0.02
        else:  # bias: conv bias (its module has a 4-d weight) -> 0 ; BN beta -> N(0,1)
            wk = k[:-len('bias')] + 'weight'
            sd[k] = torch.zeros(shp) if (wk in shapes and len(shapes[wk]) == 4) else torch.randn(shp, generator=g)
    return sd


def srgan_condition(sd):
    """In-place re-scaling of recipe values for the SRGAN nets (a name -> tensor mapping, e.g. a state_dict): the
    un-normalised VGG stack gets He-scaled conv weights (N(0, .02) through 16 plain convs underflows to zero features),
    the discriminator's Linear head N(0, .1) instead of the BatchNorm rule's ~1, and the PReLU slopes distinct positive
    values in [0.1, 0.4) (distinct, so that a swapped slope shows; gcc_prelu itself differentiates on the sign of its input and takes
    any slope: tests/test_srgan_gpu.py::test_prelu_and_pixel_shuffle runs 0 and -0.2)."""
    i = 0
    with torch.no_grad():
        for k, v in sd.items():
            if k.startswith('truncated_vgg19.') and v.dim() == 4:
                v.mul_((2.0 / (v.shape[1] * 9)) ** 0.5 / 0.02)
            elif k == 'fc1.weight':
                v.sub_(1.0).mul_(5.0)
            elif tuple(v.shape) == (1,) and k.endswith('.weight'):
                v.fill_(0.1 + 0.3 * ((i * 0.37) % 1.0))
                i += 1


def shape_bn_scales_for_removal(sd, seed):
    """In-place on a name -> tensor mapping of a full (num_downs 8) U-Net: BatchNorm scales laid out so that a magnitude
    threshold removes whole inner blocks (models/Pix2Pix.py:87, 97) -- the innermost up norm in [.50, .55], the depth-6 norms
    in [.60, .80], the depth-5 norms in [.75, .95], every other norm spread over [.30, 1.10] (so that the MAC budget falls steadily
    with the threshold and a budget search has something to find).  Thresholds in (.55, .60) remove block 7, in (.80, .95) blocks 6
    and 7, above .95 blocks 5, 6 and 7."""
    p5 = 'model.model.1' + '.model.3' * 4
    ranges = {p5 + '.model.3.model.3.model.4': (0.50, 0.55), p5 + '.model.3.model.2': (0.60, 0.80), p5 + '.model.3.model.6': (0.60, 0.80),
              p5 + '.model.2': (0.75, 0.95), p5 + '.model.6': (0.75, 0.95)}
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in sd.items():
            if v.dim() == 1 and k.endswith('.weight'):
                lo, hi = ranges.get(k[:-len('.weight')], (0.30, 1.10))
                v.copy_(lo + (hi - lo) * torch.rand(v.shape, generator=g))


def recipe_transform(cout, cin, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((cout, cin, 1, 1), generator=g) * 2 - 1) / (cin ** 0.5)


def sample_idx(n, k=2048):
    """fixed subsample positions used to store large final tensors compactly"""
    if n <= k:
        return torch.arange(n)
    return (torch.arange(k, dtype=torch.float64) * (n - 1) / (k - 1)).round().long()
