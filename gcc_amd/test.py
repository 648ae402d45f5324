"""The reference's test.py (:12-155) on the MI355X path: rebuild the (pruned) model from a checkpoint's cfg, run the
evaluation split through the generator in eval mode and write the images the reference's evaluators read
(<checkpoints_dir>/<name>/test_results/...).  The image files are read by gcc_amd.data (host decode, GPU transforms);
GCC_HOST_DATALOADER=1 takes the reference's ``data`` package from the import path instead.

    python -m gcc_amd.test --dataroot ./database/cityscapes/ --model pix2pix --pretrain_path <ckpt.pth> --name <exp>
"""
import copy
import os

import torch

from .models import get_model_class
from .options import options
from .utils import util


def _dataset(opt):
    if os.environ.get('GCC_HOST_DATALOADER') != '1':
        from .data import create_dataset
        return create_dataset(opt)
    from data import create_dataset          # the reference's loaders
    return create_dataset(opt)


def _run(model, opt, dataset, result_dir, forward, limit=None):
    util.mkdirs(result_dir)
    for i, data in enumerate(dataset):
        if limit is not None and i == limit:
            break
        model.set_input(data)
        forward()
        util.save_images(model.get_current_visuals(), model.image_paths, result_dir, direction=opt.direction,
                         aspect_ratio=opt.aspect_ratio)


def test(model, opt):
    """test.py:12-118"""
    opt = copy.deepcopy(opt)
    result_dir = os.path.join(opt.checkpoints_dir, opt.name, 'test_results')
    opt.num_threads, opt.batch_size, opt.serial_batches = 0, 1, True
    model.model_eval()
    if opt.model == 'pix2pix':
        opt.phase, opt.no_flip, opt.load_size = 'val', True, 256
        _run(model, opt, _dataset(opt), result_dir, model.forward)
    elif opt.model == 'cyclegan':
        opt.phase, opt.no_flip, opt.load_size = 'test', True, 256
        model.visual_names = ['real_A', 'fake_B']
        _run(model, opt, _dataset(opt), result_dir, model.visual_forward)
    elif opt.model == 'sagan':
        opt.load_size = 64
        _run(model, opt, _dataset(opt), result_dir, model.forward, limit=1000)
    elif opt.model == 'srgan':
        for name in ('Set5', 'Set14', 'B100', 'Urban100'):
            opt.phase = 'test/' + name
            _run(model, opt, _dataset(opt), os.path.join(result_dir, name), model.forward)
    else:
        raise NotImplementedError('%s not implemented' % opt.model)


def main(argv=None):
    opt = options.parse(argv)
    opt.isTrain = True
    util.mkdirs(os.path.join(opt.checkpoints_dir, opt.name))
    if opt.pretrain_path is None or not os.path.exists(opt.pretrain_path):
        raise FileNotFoundError('pretrain model path must be exist!!!')
    filter_cfgs, channel_cfgs = torch.load(opt.pretrain_path, map_location='cpu')['cfg']
    cls = get_model_class(opt)
    if opt.model == 'cyclegan':
        model = cls(opt, cfg_AtoB=filter_cfgs, cfg_BtoA=channel_cfgs)
    elif opt.model in ('srgan', 'sagan'):
        model = cls(opt, filter_cfgs=filter_cfgs)
    else:
        model = cls(opt, filter_cfgs=filter_cfgs, channel_cfgs=channel_cfgs)
    model.load_models(opt.pretrain_path, load_discriminator=False)
    test(model, opt)
    return model


if __name__ == '__main__':
    main()
