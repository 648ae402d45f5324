import os, sys
from collections import OrderedDict
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_pix2pix_gpu import build_model, load_recipe
from oracle import gcc_oracle as O
for D, S in ((5, 64), (6, 64), (7, 128), (8, 256)):
    for train in (False, True):
        model, _, opt = build_model(['--dataroot', './database/cityscapes/', '--model', 'pix2pix', '--gpu_ids', '0', '--ngf', '8',
                                     '--ndf', '8', '--num_downs', str(D), '--no_dropout'])
        load_recipe(model.netG, 4001)
        model.refresh_weights()
        A = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(1)) * 2 - 1
        model.model_train() if train else model.model_eval()
        model.set_input({'A': A, 'B': A.clone(), 'A_paths': ['a'] * 2, 'B_paths': ['b'] * 2})
        sdG = OrderedDict((k, v.detach().float().cpu().clone()) for k, v in model.netG.state_dict().items())
        model.forward()
        O.EMULATE_BF16 = True
        emu = O.unet_forward(sdG, A, num_downs=D, train=train)
        O.EMULATE_BF16 = False
        ref = O.unet_forward(sdG, A, num_downs=D, train=train)
        e = (model.fake_B.cpu() - emu).abs(); f = (model.fake_B.cpu() - ref).abs()
        print('D=%d %s: vs emulated max %.4g mean %.4g | vs fp32 max %.4g mean %.4g' % (D, 'train' if train else 'eval', e.max(), e.mean(), f.max(), f.mean()), flush=True)
