"""Training entry point: the reference's train.py loop (train.py:75-174) on the MI355X path.

    python -m gcc_amd.train --dataroot ./database/cityscapes/ --model pix2pix --ngf 32 --ndf 128 \\
        --online_distillation --darts_discriminator --lambda_content 50 --lambda_gram 1e4 ...
    (data parallel: python -m torch.distributed.run --nproc-per-node 8 -m gcc_amd.train ...)

Batches come from gcc_amd.data (host decode + the GPU transform chain of SURVEY.md 8(f).4) or, with
``--dataroot synthetic[:N]``, from seeded U[-1,1) pairs of the same shape.  Any iterable of the reference's
batch dicts ({'A','B','A_paths','B_paths'} ...) can be passed to main(datasets=...) instead.
The per-epoch evaluation of the reference's loop (train.py:14-73, 160-165) needs third-party evaluator networks
(Inception / DRN weights): main(evaluate=fn) takes the callable that produces the metric(s) -- gcc_amd.metric supplies
the arithmetic downstream of those networks -- and keeps the reference's best-checkpoint bookkeeping around it.
"""
import copy
import os
import time

import torch

from . import dist as gdist
from .models import get_model_class
from .options import options
from .utils import util


class SyntheticPairs:
    def __init__(self, opt, n_batches, seed):
        self.opt, self.n, self.seed = opt, n_batches, seed

    def __len__(self):
        return self.n * self.opt.batch_size

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        b, s = self.opt.batch_size, self.opt.crop_size
        img = lambda size: torch.rand(b, 3, size, size, generator=g) * 2 - 1
        for _ in range(self.n):
            if self.opt.model == 'sagan':            # data/sa_dataset.py:45
                yield {'z': torch.randn(b, self.opt.z_dim, generator=g), 'real_img': img(s), 'img_path': [''] * b}
            elif self.opt.model == 'srgan':          # data/sr_dataset.py:173
                hr = self.opt.image_size
                yield {'lr': img(hr // self.opt.upscale_factor), 'hr': img(hr), 'lr_names': [''] * b, 'hr_names': [''] * b}
            else:
                yield {'A': img(s), 'B': img(s), 'A_paths': [''] * b, 'B_paths': [''] * b}


def make_datasets(opt):
    root = str(opt.dataroot)
    if root.startswith('synthetic'):
        n = int(root.split(':')[1]) if ':' in root else 8
        r = gdist.rank()
        return SyntheticPairs(opt, n, 1234 + r), SyntheticPairs(opt, n, 4321 + r)
    # host decode + GPU transforms (gcc_amd.data); two loaders over the same phase, as create_split_dataset builds
    # them (data/__init__.py:59-66)
    from .data import create_dataset
    return create_dataset(opt), create_dataset(opt)


def attach_teacher(model, opt, model_class):
    """train.py:92-105"""
    topt = copy.deepcopy(opt)
    topt.ngf, topt.ndf = opt.teacher_ngf, opt.teacher_ndf
    topt.darts_discriminator = topt.online_distillation = False
    topt.generator_only = False
    teacher = model_class(topt)
    teacher.model_train()
    if opt.teacher_initial_path is not None:
        teacher.load_models(opt.teacher_initial_path, load_discriminator=False)
    model.teacher_model = teacher
    model.init_distillation()
    teacher.init_distillation()
    return teacher


class BestRecord:
    """utils/best_information.py:1-33 -- best value and epoch per metric slot.  Larger is better for the SRGAN scores and the
    cityscapes mIoU, smaller for every FID; ties count as an improvement (the reference compares with <= / >=)."""

    def __init__(self, opt):
        self.higher = opt.model == 'srgan' or 'cityscapes' in str(opt.dataroot)
        self.value, self.epoch = {}, {}

    def update(self, metric, epoch, index=0):
        best = self.value.get(index, 0.0 if self.higher else float('inf'))
        if (best <= metric) if self.higher else (best >= metric):
            self.value[index], self.epoch[index] = metric, epoch
            return True
        return False

    def report(self, logger, last):
        last = list(last) if isinstance(last, (list, tuple)) else [last]
        logger.info(' | '.join('slot %d: best epoch %d %.2f / last %.2f' % (i, self.epoch.get(i, 0), self.value.get(i, float('nan')), v)
                               for i, v in enumerate(last)))


def run_evaluation(model, opt, logger, epoch, best, evaluate, ckpt_dir):
    """train.py:14-73: evaluate(model, opt) returns [(metric value, direction tag), ...] in the reference's slot order
    (pix2pix: one mIoU or FID tagged opt.direction; cyclegan: AtoB, BtoA FIDs; sagan: one FID; srgan: 4 PSNR then 4 SSIM
    tagged with the test-set names); a new best in a slot saves model_best_<tag>.pth on rank 0."""
    model.model_eval()
    scores = evaluate(model, copy.deepcopy(opt))
    model.model_train()
    for i, (value, tag) in enumerate(scores):
        logger.info('evaluation slot %d (%s): %.2f' % (i, tag, value))
        if best.update(value, epoch, index=i):
            model.save_models(epoch, ckpt_dir, fid=value, isbest=True, direction=tag)
    return [v for v, _ in scores]


def main(argv=None, datasets=None, evaluate=None):
    gdist.init_from_env()
    opt = options.parse(argv)
    opt.isTrain = True
    exp = os.path.join(opt.checkpoints_dir, opt.name)
    util.mkdirs(exp)
    logger = util.get_logger(os.path.join(exp, 'logger.log' if gdist.rank() == 0 else 'logger.rank%d.log' % gdist.rank()))
    model_class = get_model_class(opt)
    model = model_class(opt)
    if opt.norm_prune or opt.scale_prune:
        from .utils.prune_util import cyclegan_prune, prune
        model = cyclegan_prune(model, opt, logger) if 'cyclegan' in opt.model else prune(model, opt, logger)
    if opt.online_distillation:
        attach_teacher(model, opt, model_class)
    if opt.initial_path is not None:
        model.load_models(opt.initial_path, load_discriminator=False)
    train_set, val_set = datasets if datasets is not None else make_datasets(opt)
    logger.info('The number of training images = %d' % len(train_set))
    total_iters = 0
    best, last_scores = BestRecord(opt), None
    # GCC_REPLAY=1: the iteration is recorded once per epoch and re-issued from native code (gcc_amd.replay: the launch-bound
    # models -- CycleGAN at batch 1, SAGAN, SRGAN -- train at the GPU's pace instead of the Python host's); same results
    replay = None
    if os.environ.get('GCC_REPLAY', '0') == '1':
        from .replay import IterationReplay
        replay = IterationReplay(model, opt, enabled=True)
    for epoch in range(opt.epoch_count, opt.n_epochs + opt.n_epochs_decay + 1):
        model.model_train()
        logger.info('\nEpoch:%d' % epoch)
        t_epoch = time.time()
        val_iter = iter(val_set)
        epoch_iter = 0
        for data in train_set:
            t0 = time.time()
            total_iters += opt.batch_size
            epoch_iter += opt.batch_size
            if replay is not None:
                arch = opt.darts_discriminator and model.teacher_model is not None
                replay.step(data, next(val_iter) if arch else None)
            else:
                model.set_input(data)
                model.optimize_parameters()
                if opt.darts_discriminator and model.teacher_model is not None:
                    model.set_input(next(val_iter))
                    model.clipping_mask_alpha()
                    model.optimizer_netD_arch()
            if total_iters % opt.print_freq == 0:
                losses = model.get_current_losses()
                msg = '(epoch: %d, iters: %d, time: %.3f) ' % (epoch, epoch_iter, (time.time() - t0) / opt.batch_size)
                logger.info(msg + ' '.join('%s: %.3f' % kv for kv in losses.items()))
        if epoch % opt.save_epoch_freq == 0:            # train.py:160-165
            if evaluate is not None:
                last_scores = run_evaluation(model, opt, logger, epoch, best, evaluate, os.path.join(exp, 'checkpoints'))
            logger.info('saving the model at the end of epoch %d, iters %d' % (epoch, total_iters))
            if epoch == opt.n_epochs + opt.n_epochs_decay:
                model.save_models(epoch, os.path.join(exp, 'checkpoints'))
        model.print_sparse_info(logger)
        logger.info('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.n_epochs + opt.n_epochs_decay,
                                                                    time.time() - t_epoch))
        model.update_learning_rate(epoch)
        if replay is not None:
            replay.invalidate()          # learning rates (and the EMA beta) are launch arguments of the recording
    if last_scores is not None:
        best.report(logger, last_scores)
    return model


if __name__ == '__main__':
    main()
