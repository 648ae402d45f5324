"""Flag surface of the reference (options/options.py:6-126) and its per-model post-processing
(parse(), :154-228), table-driven.  Every flag name, type and default is preserved so the
reference's scripts/*.sh drive this train.py unchanged.  Data-parallel execution is configured by
the launcher environment (WORLD_SIZE / RANK / LOCAL_RANK), not by new mandatory flags."""
import argparse

INF = float('inf')

# (name, type-or-'flag', default, help)
_FLAGS = [
    # basic
    ('dataroot', str, None, 'path to images'),
    ('name', str, 'default', 'experiment name'),
    ('gpu_ids', str, '0', 'gpu ids: e.g. 0  0,1,2  use -1 for CPU (rank-local device comes from LOCAL_RANK under DP)'),
    ('checkpoints_dir', str, './experiments', 'models are saved here'),
    ('phase', str, 'train', 'train, val, test'),
    ('load_path', str, None, 'unused by the reference'),
    ('pretrain_path', str, None, 'pretrained model to prune'),
    # model
    ('model', str, 'pix2pix', '[cyclegan | pix2pix | sagan | srgan]'),
    ('input_nc', int, 3, ''), ('output_nc', int, 3, ''),
    ('ngf', int, 64, ''), ('pretrain_ngf', int, 64, ''), ('ndf', int, 128, ''),
    ('backbone', str, 'unet', '[unet | resnet]'),
    ('no_dropout', 'flag', False, 'no dropout for the generator'),
    ('num_downs', int, 8, ''),
    ('continue_train', bool, False, 'unused by the reference'),
    # dataset
    ('dataset_mode', str, 'aligned', ''), ('direction', str, 'AtoB', 'AtoB or BtoA'),
    ('serial_batches', 'flag', False, ''), ('num_threads', int, 8, ''), ('batch_size', int, 1, ''),
    ('load_size', int, 286, ''), ('crop_size', int, 256, ''), ('max_dataset_size', int, INF, ''),
    ('preprocess', str, 'resize_and_crop', ''), ('no_flip', 'flag', False, ''), ('split_dataset', 'flag', False, ''),
    # train
    ('print_freq', int, 500, ''), ('save_epoch_freq', int, 1, ''), ('epoch_count', int, 1, ''),
    ('n_epochs', int, 100, ''), ('n_epochs_decay', int, 150, ''), ('lr', float, 0.0002, ''),
    ('gan_mode', str, 'hinge', '[vanilla | lsgan | hinge | wgangp]'), ('pool_size', int, 100, ''),
    ('lr_policy', str, 'linear', '[linear | step | plateau | cosine]'), ('lr_decay_iters', int, 50, ''),
    ('lambda_A', float, 10.0, ''), ('lambda_B', float, 10.0, ''), ('lambda_identity', float, 0.5, ''),
    ('lambda_L1', float, 0.0, ''),
    # test
    ('ntest', int, INF, ''), ('aspect_ratio', float, 1.0, ''),
    ('drn_path', str, './database/cityscapes/drn-d-105_ms_cityscapes.pth', ''),
    # prune
    ('scale_prune', 'flag', False, ''), ('norm_prune', 'flag', False, ''),
    ('lambda_weight', float, 0.0, ''), ('lambda_scale', float, 0.0, ''),
    ('target_budget', float, None, ''), ('target_budget_B', float, None, ''), ('lottery_path', str, None, ''),
    # darts
    ('darts_discriminator', 'flag', False, ''), ('arch_lr', float, 1e-4, ''), ('arch_lr_step', 'flag', False, ''),
    ('lambda_alpha', float, 0.01, ''), ('ema_beta', float, 1.0, ''), ('adaptive_ema', 'flag', False, ''),
    ('regular', 'flag', False, ''), ('arch_base_loss', 'flag', False, ''), ('only_arch_base', 'flag', False, ''),
    ('normalize_arch', 'flag', False, ''), ('clear_arch', 'flag', False, ''), ('threshold', float, 0.5, ''),
    # distillation
    ('online_distillation', 'flag', False, ''), ('normal_distillation', 'flag', False, ''),
    ('distillation_path', str, None, ''), ('lambda_content', float, 0.0, ''), ('lambda_gram', float, 0.0, ''),
    ('teacher_ngf', int, 64, ''), ('teacher_ndf', int, 64, ''),
    # super-resolution
    ('lambda_SR_adversarial', float, 1e-3, ''), ('lambda_SR_content', float, 0.0, ''),
    ('lambda_SR_perceptual', float, 1, ''), ('image_size', int, 96, ''), ('upscale_factor', int, 4, ''),
    ('lr_img_type', str, 'imagenet-norm', ''), ('hr_img_type', str, '[-1, 1]', ''),
    ('initial_path', str, None, ''), ('teacher_initial_path', str, None, ''),
    # noise gan
    ('z_dim', int, 128, ''), ('center_crop', 'flag', False, ''),
    # declared here because the reference reads it (options.py:196) without declaring it
    ('generator_only', 'flag', False, 'SRGAN generator-only pre-training'),
]

parser = argparse.ArgumentParser('GAN-Compression')
for _n, _t, _d, _h in _FLAGS:
    if _t == 'flag':
        parser.add_argument('--' + _n, action='store_true', help=_h)
    else:
        parser.add_argument('--' + _n, type=_t, default=_d, help=_h)


def postprocess(opt):
    """Per-model overrides of parse() (options/options.py:156-228)."""
    opt.gpu_ids = [i for i in (int(s) for s in str(opt.gpu_ids).split(',')) if i >= 0] \
        if not isinstance(opt.gpu_ids, list) else opt.gpu_ids
    root = opt.dataroot or ''
    if opt.model in ('pix2pix', 'newpix2pix'):
        opt.norm, opt.dataset_mode, opt.no_flip, opt.load_size = 'batch', 'aligned', True, 256
        opt.pool_size, opt.teacher_ndf, opt.lambda_L1 = 0, 128, 100.0
        if 'cityscapes' in root:
            opt.direction, opt.save_epoch_freq, opt.n_epochs, opt.n_epochs_decay, opt.print_freq = 'BtoA', 5, 100, 150, 100
        if 'edges2shoes' in root:
            opt.batch_size, opt.n_epochs, opt.n_epochs_decay = 4, 10, 30
        if 'maps' in root:
            opt.n_epochs, opt.direction, opt.no_flip, opt.load_size = 100, 'BtoA', False, 286
            opt.n_epochs_decay, opt.save_epoch_freq, opt.print_freq, opt.lambda_L1 = 200, 5, 100, 10.0
    elif opt.model == 'srgan':
        opt.dataset_mode, opt.gan_mode, opt.lr = 'sr', 'vanilla', 1e-4
        opt.n_epochs_decay, opt.batch_size = 0, 16
        if opt.generator_only:
            opt.n_epochs = 130
        else:
            opt.n_epochs, opt.lr_policy = 30, 'step'
            opt.lr_decay_iters = opt.n_epochs // 2
    elif opt.model == 'sagan':
        opt.dataset_mode, opt.crop_size, opt.batch_size, opt.lr = 'sa', 64, 64, 1e-4
        opt.n_epochs_decay, opt.save_epoch_freq = 0, 5
        opt.n_epochs, opt.center_crop = (300, False) if 'church' in root else (100, True)
    elif 'cyclegan' in opt.model:
        opt.dataset_mode, opt.gan_mode, opt.n_epochs, opt.n_epochs_decay, opt.print_freq = 'unaligned', 'lsgan', 100, 100, 100
    if opt.lambda_weight > 0 or opt.lambda_scale > 0:
        opt.n_epochs //= 10
        opt.n_epochs_decay //= 10
    return opt


def parse(argv=None):
    return postprocess(parser.parse_args(argv))
