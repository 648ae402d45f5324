"""Checkpoint interchange with the reference (models/Pix2Pix.py:636-658): a file written by the reference's
save_models loads into the HIP model and reproduces the reference's eval image; what the HIP model saves has the
reference's layout (dict keys, state_dict key order, NCHW-contiguous fp32 CPU tensors) and, without a step in between,
the same bits."""
import os

import numpy as np
import pytest
import torch

from tests.test_pix2pix_gpu import DEV  # noqa: F401

pytestmark = pytest.mark.gpu

ARGV = ['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '0', '--ngf', '4', '--ndf', '4',
        '--num_downs', '6', '--load_size', '64', '--crop_size', '64', '--darts_discriminator']


def test_reference_checkpoint_round_trip(golden_dir, tmp_path):
    from gcc_amd.options import options
    from gcc_amd.models import get_model_class
    ref_path = os.path.join(golden_dir, 'ref_checkpoint_pix2pix.pth')
    z = np.load(os.path.join(golden_dir, 'ref_checkpoint_pix2pix.npz'))
    opt = options.parse(ARGV)
    opt.isTrain = True
    model = get_model_class(opt)(opt)
    fid, best = model.load_models(ref_path)
    assert fid == 12.5 and best == float('inf')
    model.model_eval()
    A = torch.from_numpy(z['A'])
    model.set_input({'A': A, 'B': A.clone(), 'A_paths': ['a'], 'B_paths': ['b']})
    model.forward()
    e = (model.fake_B.cpu() - torch.from_numpy(z['fake_B'])).abs()
    assert e.max() <= 2e-2 and e.mean() <= 3e-3, (float(e.max()), float(e.mean()))
    model.save_models(7, str(tmp_path), fid=12.5)
    ours = torch.load(os.path.join(str(tmp_path), 'model_7.pth'), map_location='cpu')
    ref = torch.load(ref_path, map_location='cpu')
    assert list(ours.keys()) == list(ref.keys()) == ['G', 'D', 'epoch', 'cfg', 'fid']
    assert ours['epoch'] == ref['epoch'] == 7 and ours['cfg'] == ref['cfg'] and ours['fid'] == ref['fid']
    for part in ('G', 'D'):
        assert list(ours[part].keys()) == list(ref[part].keys())
        for k, v in ref[part].items():
            o = ours[part][k]
            assert o.dtype == v.dtype and o.shape == v.shape and o.is_contiguous() and o.device.type == 'cpu', (part, k)
            assert torch.equal(o, v), (part, k)
    # best-model naming (:645-648)
    model.save_models(9, str(tmp_path), fid=3.0, isbest=True, direction='BtoA')
    assert os.path.exists(os.path.join(str(tmp_path), 'model_best_BtoA.pth'))
    # generator-only load (train.py passes load_discriminator=False for --initial_path / --pretrain_path)
    fresh = get_model_class(opt)(opt)
    d_before = {k: v.clone() for k, v in fresh.netD.state_dict().items()}
    fresh.load_models(ref_path, load_discriminator=False)
    assert all(torch.equal(v, fresh.netD.state_dict()[k]) for k, v in d_before.items())
    assert all(torch.equal(v.cpu(), ref['G'][k]) for k, v in fresh.netG.state_dict().items())


OTHER = {
    'cyclegan': (['--dataroot', './database/horse2zebra/', '--model', 'cyclegan', '--gpu_ids', '0', '--ngf', '8', '--ndf', '8',
                  '--teacher_ngf', '16', '--darts_discriminator', '--lambda_content', '0.01', '--lambda_gram', '10',
                  '--arch_lr', '1e-4', '--arch_lr_step'], ['G_A', 'G_B', 'D_A', 'D_B'], 'fid'),
    'sagan': (['--dataroot', './database/celeb/', '--model', 'sagan', '--gpu_ids', '0', '--ngf', '8', '--ndf', '8',
               '--darts_discriminator', '--threshold', '0.1'], ['G', 'D'], 'fid'),
    'srgan': (['--dataroot', './database/sr/', '--model', 'srgan', '--gpu_ids', '0', '--ngf', '8', '--ndf', '8',
               '--darts_discriminator'], ['G', 'D'], 'psnr'),
}


@pytest.mark.parametrize('which', ['cyclegan', 'sagan', 'srgan'])
def test_reference_checkpoint_other_models(which, golden_dir, tmp_path):
    """models/CycleGAN.py:654-680, models/SAGAN.py:578-600, models/SRGAN.py:578-600: a reference-written file loads, the
    eval image matches the reference's, and the file written back has the reference's layout and, untouched, its bits"""
    from gcc_amd.options import options
    from gcc_amd.models import get_model_class
    if which == 'srgan':
        os.environ.setdefault('GCC_VGG19_RANDOM', '1')      # the perceptual network is not part of the checkpoint
    argv, parts, score_key = OTHER[which]
    ref_path = os.path.join(golden_dir, 'ref_checkpoint_%s.pth' % which)
    z = np.load(os.path.join(golden_dir, 'ref_checkpoint_%s.npz' % which))
    opt = options.parse(argv)
    opt.isTrain = True
    model = get_model_class(opt)(opt)
    model.load_models(ref_path)
    model.model_eval()
    if which == 'cyclegan':
        model.set_input({'A': torch.from_numpy(z['A']), 'B': torch.from_numpy(z['B']), 'A_paths': ['a'], 'B_paths': ['b']})
        model.forward()
        pairs = [(model.fake_B, z['fake_B']), (model.fake_A, z['fake_A'])]
    elif which == 'sagan':
        model.set_input({'z': torch.from_numpy(z['z']), 'real_img': torch.zeros(4, 3, 64, 64), 'img_path': ['p'] * 4})
        model.forward()
        pairs = [(model.fake_img, z['fake_img'])]
    else:
        model.set_input({'lr': torch.from_numpy(z['lr']), 'hr': torch.zeros(2, 3, 48, 48), 'lr_names': ['a'] * 2, 'hr_names': ['b'] * 2})
        model.forward()
        pairs = [(model.fake_hr, z['fake_hr'])]
    for got, ref in pairs:
        e = (got.float().cpu() - torch.from_numpy(ref)).abs()
        # image tolerance of the bf16 path against the fp32 reference: the bound of the two-iteration golden tests of these
        # models (tests/test_cyclegan_gpu.py: max 4e-2, mean 6.25e-3 on [-1, 1] images)
        assert e.max() <= 4e-2 and e.mean() <= 6.25e-3, (which, float(e.max()), float(e.mean()))
    fresh = get_model_class(opt)(opt)
    fresh.load_models(ref_path)
    fresh.save_models(5, str(tmp_path), fid=7.25)
    ours = torch.load(os.path.join(str(tmp_path), 'model_5.pth'), map_location='cpu')
    ref = torch.load(ref_path, map_location='cpu')
    assert list(ours.keys()) == list(ref.keys()) == parts + ['epoch', 'cfg', score_key]
    assert ours['epoch'] == ref['epoch'] == 5 and ours[score_key] == ref[score_key] == 7.25 and ours['cfg'] == ref['cfg']
    for part in parts:
        assert list(ours[part].keys()) == list(ref[part].keys()), part
        for k, v in ref[part].items():
            o = ours[part][k]
            assert o.dtype == v.dtype and o.shape == v.shape and o.device.type == 'cpu', (part, k)
            assert torch.equal(o, v), (part, k)
