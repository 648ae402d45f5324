"""One-shot pruning to a MAC budget (reference utils/prune_util.py:6-63).

The budget is measured with the ``thop`` convention the reference's scripts were tuned against
(thop is an un-pinned, un-vendored dependency of the reference, so the convention is restated here and
pinned only by the indirect known answers of SURVEY.md section 4: U-Net ngf 64 -> 18.14 G conv MACs):
every Conv2d *and* ConvTranspose2d counts ``out_elements * (Cin / groups) * kh * kw`` multiply-
accumulates, BatchNorm2d counts ``2 * elements``; activations, dropout and Tanh count nothing.
The count is pure shape arithmetic on the generator's module tree -- no forward pass is needed.
"""
import torch
import torch.nn as nn


def unet_macs(netG, size=256):
    """(MACs in G, params in M) of a UnetGenertor parameter tree for a 1 x 3 x size x size input."""
    convs = [m for m in netG.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
    downs = [m for m in convs if isinstance(m, nn.Conv2d)]
    ups = [m for m in convs if isinstance(m, nn.ConvTranspose2d)]
    total = 0
    h = size
    hs = []
    for m in downs:                       # named_modules order: outermost -> innermost
        h //= 2
        hs.append(h)
        total += h * h * m.out_channels * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
    for m in ups:                         # innermost -> outermost
        h *= 2
        total += h * h * m.out_channels * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
    # BatchNorm: 2 ops per element of the tensor it normalises
    D = len(downs)
    bns = [m for m in netG.modules() if isinstance(m, nn.BatchNorm2d)]
    for i, m in enumerate(bns):
        if i < D - 2:                     # down norms at depth 1..D-2
            hh = hs[i + 1]
        else:                             # up norms, innermost first: output of up conv at depth D-1, D-2, ...
            hh = hs[D - 2 - (i - (D - 2))]
        total += 2 * m.num_features * hh * hh
    params = sum(p.numel() for p in netG.parameters())
    return total / 1000 ** 3, params / 1000 ** 2


def get_flops_parms(model_netG, device, opt, verbose=False):
    return unet_macs(model_netG, opt.load_size)


def _prefix(d):
    return 'model' if d == 0 else 'model.model.1' + '.model.3' * (d - 1)


def bn_names(num_downs=8):
    """BatchNorm2d modules of the U-Net in the reference's named_modules() order"""
    D = num_downs
    return [_prefix(d) + '.model.2' for d in range(1, D - 1)] + [_prefix(D - 1) + '.model.4'] + \
           [_prefix(d) + '.model.6' for d in range(D - 2, 0, -1)]


def scale_prune_cfg(sd, threshold, ngf, num_downs=8):
    """filter_cfgs / channel_cfgs of models/Pix2Pix.py:823-860: count of gamma > tau per BatchNorm (raw gamma,
    no abs) with the reference's zero-propagation rules.  sd: name -> CPU tensor of the BN weights."""
    D = num_downs
    f, c = [ngf], [ngf]
    inner_up = _prefix(D - 1) + '.model.4'
    last_down = _prefix(D - 2) + '.model.2'
    up_flag, up_num = False, 0
    for name in bn_names(D):
        cnt = int((sd[name + '.weight'] > threshold).sum())
        f.append(cnt)
        if name == inner_up:
            up_flag = True
            if cnt == 0:
                f[-2] = 0
        if up_flag:
            up_num += 1
            if f[-2 * up_num] == 0:
                f[-1] = 0
                cnt = 0
            c.append(cnt + f[-1 - 2 * up_num])
        else:
            c.append(cnt)
        if name == last_down:
            zero = f[-1] == 0
            f.append(0 if zero else ngf * 8)
            c.append(0 if zero else ngf * 8)
    return f, c


def max_min_bn_scale(sd, num_downs=8):
    """models/Pix2Pix.py:754-776 as 0-dim fp32 tensors (what the reference's search iterates on)"""
    p3 = _prefix(5)
    prunable = [p3 + '.model.2', p3 + '.model.3.model.2', p3 + '.model.3.model.3.model.4',
                p3 + '.model.3.model.6', p3 + '.model.6']
    un_max, pr_max, mn = None, None, None
    for name in bn_names(num_downs):
        w = sd[name + '.weight']
        if name in prunable:
            pr_max = w.max() if pr_max is None else torch.max(w.max(), pr_max)
        else:
            un_max = w.max() if un_max is None else torch.min(w.max(), un_max)
        mn = w.min() if mn is None else torch.min(w.min(), mn)
    return torch.min(pr_max, un_max), mn


def binarysearch_threshold_sd(sd, opt, target_budget):
    """utils/prune_util.py:20-47 on a state_dict -- the interval end points and the mid point stay 0-dim fp32
    tensors, as in the reference, so the thresholds (and therefore the integer cfgs) are reproduced bit for bit."""
    if not opt.scale_prune:
        raise NotImplementedError('norm pruning is not on the MI355X path yet')
    max_scale, min_scale = max_min_bn_scale(sd, opt.num_downs)
    root = str(opt.dataroot)
    tolerance = 0.01 if 'sr' in root else (0.001 if ('celeb' in root or 'church' in root) else 0.1)
    while max_scale > min_scale:
        mid = (max_scale + min_scale) / 2
        f, c = scale_prune_cfg(sd, mid, opt.ngf, opt.num_downs)
        budget = cfg_macs(opt, f, c)
        if abs(target_budget - budget) <= tolerance:
            return mid
        elif target_budget - budget > tolerance:
            max_scale = mid
        else:
            min_scale = mid
    raise NotImplementedError('No appropriate threshold found')


def binarysearch_threshold(model, target_budget):
    sd = {k: v.detach().cpu() for k, v in model.netG.state_dict().items() if k.endswith('.weight') and v.dim() == 1}
    return binarysearch_threshold_sd(sd, model.opt, target_budget)


def cfg_macs(opt, f, c):
    """thop-convention G-MACs of UnetGenertor(filter_cfgs=f, channel_cfgs=c) without building it
    (a block whose widths are zero is absent, models/Pix2Pix.py:87,97)."""
    size = opt.load_size
    k2 = 16
    total = 0
    D = 8
    h = [size >> (d + 1) for d in range(D)]
    present = [True] * D
    present[7] = f[7] != 0 and f[8] != 0
    for i in range(3):
        present[6 - i] = f[6 - i] != 0 and f[9 + i] != 0
    down_in = [3] + list(c[:7])
    up_in = [c[14 - d] for d in range(D)]
    up_out = [3] + [f[15 - d] for d in range(1, D)]
    for d in range(D):
        if not present[d]:
            continue
        total += h[d] * h[d] * f[d] * down_in[d] * k2
        hh = h[d] * 2
        total += hh * hh * up_out[d] * up_in[d] * k2
        if 0 < d < D - 1:
            total += 2 * f[d] * h[d] * h[d]
        if d > 0:
            total += 2 * up_out[d] * hh * hh
    return total / 1000 ** 3


def prune(model, opt, logger):
    """utils/prune_util.py:49-63"""
    if opt.target_budget is None:
        raise NotImplementedError('the target budget must be exist!!!')
    if opt.pretrain_path is None:
        raise NotImplementedError('the pretrain path must be exist!!!')
    model.load_models(opt.pretrain_path, load_discriminator=False)
    threshold = binarysearch_threshold(model, opt.target_budget)
    pruned_model = model.prune(threshold, lottery_path=getattr(opt, 'lottery_path', None))
    filter_cfg, channel_cfg = pruned_model.get_cfg()
    macs, params = get_flops_parms(pruned_model.netG, pruned_model.device, pruned_model.opt)
    logger.info(filter_cfg)
    logger.info(channel_cfg)
    logger.info('MACs:%.7f G  |  Params:%.4f M' % (macs, params))
    return pruned_model
