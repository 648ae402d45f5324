"""gan_loss_kernel on the headline configuration's 16 x 30 x 30 PatchGAN map: us per launch (value + gradient), hinge"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gcc_amd import ops
DEV = torch.device('cuda:0')
pred = ops.new_act(16, 1, 30, 30, DEV); pred.normal_()
dp = ops.new_act(16, 1, 30, 30, DEV)
loss = torch.zeros(4, device=DEV)
for _ in range(10):
    ops.gan_loss('hinge', pred, True, True, loss[0:1], dpred=dp, grad_weight=0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    ops.gan_loss('hinge', pred, True, True, loss[0:1], dpred=dp, grad_weight=0.5)
e1.record(); torch.cuda.synchronize()
print('gan_loss (hinge, value + gradient, 14400 values): %.2f us per launch back to back' % (e0.elapsed_time(e1) * 1e3 / 200))
