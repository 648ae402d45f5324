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


def tensor2imgs(image_tensor, imtype=None, normalize=True, tile=False):
    """utils/util.py:45-76: [-1, 1] (or [0, 1]) image tensors -> uint8 HWC numpy images (a 4-D batch gives [N, H, W, C]).
    The value path of the evaluators: clip((x + 1) / 2 * 255) truncated to uint8."""
    import numpy as np
    imtype = np.uint8 if imtype is None else imtype
    if isinstance(image_tensor, list):
        return [tensor2imgs(t, imtype, normalize) for t in image_tensor]
    if tile:
        raise NotImplementedError('tiled visualisation is outside the evaluation path')
    if image_tensor.dim() == 4:
        return np.concatenate([tensor2imgs(t)[None] for t in image_tensor], axis=0)
    if image_tensor.dim() == 2:
        image_tensor = image_tensor.unsqueeze(0)
    image_numpy = image_tensor.detach().cpu().float().numpy()
    if normalize:
        image_numpy = (np.transpose(image_numpy, (1, 2, 0)) + 1) / 2.0 * 255.0
    else:
        image_numpy = np.transpose(image_numpy, (1, 2, 0)) * 255.0
    image_numpy = np.clip(image_numpy, 0, 255)
    if image_numpy.shape[2] == 1:
        image_numpy = image_numpy[:, :, 0]
    return image_numpy.astype(imtype)


def tensor2im(input_image, imtype=None):
    """utils/util.py:78-95: first image of an NCHW tensor in [-1, 1] -> HWC array, (x + 1) / 2 * 255 truncated to uint8
    (no clipping, as there)"""
    import numpy as np
    import torch
    imtype = np.uint8 if imtype is None else imtype
    if isinstance(input_image, np.ndarray):
        return input_image.astype(imtype)
    if not isinstance(input_image, torch.Tensor):
        return input_image
    image_numpy = input_image.data[0].cpu().float().numpy()
    return ((np.transpose(image_numpy, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(imtype)


def save_image(image_numpy, image_path, aspect_ratio=1.0):
    """utils/util.py:153-168"""
    from PIL import Image
    image_pil = Image.fromarray(image_numpy)
    h, w, _ = image_numpy.shape
    if aspect_ratio > 1.0:
        image_pil = image_pil.resize((h, int(w * aspect_ratio)), Image.BICUBIC)
    if aspect_ratio < 1.0:
        image_pil = image_pil.resize((int(h / aspect_ratio), w), Image.BICUBIC)
    image_pil.save(image_path)


def save_images(visuals, img_path, save_image_dir, direction='AtoB', aspect_ratio=1.0, width=256):
    """utils/util.py:208-235: the input image under its partner's name, every generated image under <label>/"""
    imageA_path = img_path[0][0] if direction == 'AtoB' else img_path[1][0]
    imageB_path = img_path[1][0] if direction == 'AtoB' else img_path[0][0]
    imageA_name = (imageA_path.split('/')[-1].split('\\\\')[-1]).split('.')[0]
    imageB_name = (imageB_path.split('/')[-1].split('\\\\')[-1]).split('.')[0]
    for label, im_data in visuals.items():
        if label == 'real_A' or label == 'real_img':
            save_image(tensor2im(im_data), os.path.join(save_image_dir, imageB_name + '.png'), aspect_ratio)
        if label in ('fake_B', 'fake_A', 'fake_hr', 'fake_img'):
            image_name = '%s_%s.png' % (imageB_name if label == 'fake_A' else imageA_name, label)
            save_path = os.path.join(save_image_dir, label)
            mkdirs(save_path)
            save_image(tensor2im(im_data), os.path.join(save_path, image_name), aspect_ratio=aspect_ratio)
