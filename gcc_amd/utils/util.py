"""The three helpers of the reference's utils/util.py that sit on the training path:
init_weights (:261-286), get_scheduler (:288-303), get_logger (:246-259), plus mkdirs."""
import logging
import os

import torch.nn.init as init
from torch.optim import lr_scheduler


def mkdirs(paths):
    for p in ([paths] if isinstance(paths, str) else paths):
        os.makedirs(p, exist_ok=True)


def get_logger(file_path):
    logger = logging.getLogger('Mask-GAN')
    fmt = logging.Formatter('%(asctime)s | %(message)s', datefmt='%m/%d %I:%M:%S %p')
    for h in (logging.FileHandler(file_path), logging.StreamHandler()):
        h.setFormatter(fmt)
        logger.addHandler(h)
    logger.setLevel(logging.INFO)
    return logger


_INIT = {
    'normal': lambda w, g: init.normal_(w, 0.0, g),
    'xavier': lambda w, g: init.xavier_normal_(w, gain=g),
    'kaiming': lambda w, g: init.kaiming_normal_(w, a=0, mode='fan_in'),
    'orthogonal': lambda w, g: init.orthogonal_(w, gain=g),
}


def init_weights(net, init_type='normal', init_gain=0.02):
    """Conv*/Linear weights by init_type, their biases 0; BatchNorm2d gamma ~ N(1, gain) and
    beta ~ N(0, 1) (the reference passes no std for beta: utils/util.py:283)."""
    if init_type not in _INIT:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    for m in net.modules():
        cname = type(m).__name__
        if hasattr(m, 'weight') and ('Conv' in cname or 'Linear' in cname):
            _INIT[init_type](m.weight.data, init_gain)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in cname:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.normal_(m.bias.data, 0.0)


def lr_lambda_linear(opt):
    return lambda epoch: 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)


def get_scheduler(optimizer, opt):
    if opt.lr_policy == 'linear':
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lr_lambda_linear(opt))
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)


def tensor2imgs(image_tensor, imtype=None, normalize=True):
    """The evaluators' value path (reference utils/util.py:45-76): float image tensor(s) -> uint8 images, channels last;
    a 4-D batch gives [N, H, W, C], a list gives a list.  normalize: [-1, 1] input, (x + 1) / 2 * 255 in fp32, clipped
    to [0, 255] and truncated; otherwise [0, 1] input scaled by 255."""
    import numpy as np
    if isinstance(image_tensor, (list, tuple)):
        return [tensor2imgs(t, imtype, normalize) for t in image_tensor]
    x = image_tensor.detach().float().cpu()
    if x.dim() == 2:
        x = x[None]
    batched = x.dim() == 4
    if not batched:
        x = x[None]
    x = ((x + 1.0) / 2.0 * 255.0) if normalize else (x * 255.0)
    out = x.clamp(0.0, 255.0).permute(0, 2, 3, 1).numpy().astype(np.uint8 if imtype is None else imtype)
    if out.shape[-1] == 1:
        out = out[..., 0]
    return out if batched else out[0]
