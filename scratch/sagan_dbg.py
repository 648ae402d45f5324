import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_sagan_gpu import _build, _batch
from tests.test_pix2pix_gpu import load
from gcc_amd import ops
z = load('tests/golden', 'sagan_gcc.npz')
model, teacher, opt = _build(z)
model.model_train()
model.set_input(_batch(z, 'it0.z', 'it0.real'))
model.forward()
c = model._gctx
from oracle import gcc_oracle as O
from tests.test_oracle_golden import build_sagan_oracle
om, ot, _ = build_sagan_oracle(z)
sd = om.G
x = torch.from_numpy(z['it0.z'])
import torch.nn.functional as F
h = x.reshape(4, 128, 1, 1)
for i, (stride, pad) in enumerate(((1, 0), (2, 1), (2, 1), (2, 1)), start=1):
    w = O.spectral_weight(sd, 'l%d.0.module' % i)
    h = F.conv_transpose2d(h, w, sd['l%d.0.module.bias' % i], stride=stride, padding=pad)
    raw = ops.nhwc_to_nchw(c.raw[i - 1]).cpu()
    print('layer', i, 'raw rel', float((raw - h).norm() / h.norm()), 'sigma dev', float(c.sn[i - 1].sigma))
    h = F.relu(O.batch_norm(sd, 'l%d.1' % i, h, True))
    act = ops.nhwc_to_nchw(c.act[i - 1]).cpu()
    print('   act rel', float((act - h).norm() / h.norm()))
    if i == 3:
        h = O.self_attention(sd, 'attn1', h)
        print('   attn1 rel', float((ops.nhwc_to_nchw(c.attn[0].y).cpu() - h).norm() / h.norm()))
h = O.self_attention(sd, 'attn2', h)
print('   attn2 rel', float((ops.nhwc_to_nchw(c.attn[1].y).cpu() - h).norm() / h.norm()))
