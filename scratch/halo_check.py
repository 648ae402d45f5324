import sys, os
sys.path.insert(0, '/root/repo')
import torch
from gcc_amd import _lib, ops
lib = ops.lib()
DEV='cuda:0'
g = torch.Generator().manual_seed(1)
def run(N,H,W,Ci,Co):
    k,s,p=4,2,1
    Ho,Wo=H//2,W//2
    x = ops.new_act(N, Ci, H, W, DEV); x.copy_(torch.randn(N,Ci,H,W,generator=g).bfloat16().to(DEV))
    dy = ops.new_act(N, Co, Ho, Wo, DEV); dy.copy_(torch.randn(N,Co,Ho,Wo,generator=g).bfloat16().to(DEV))
    m = (torch.randn(Co,Ci,k,k,generator=g)*0.05).to(DEV).contiguous(memory_format=torch.channels_last)
    w, wt = ops.pack_weights(m)
    outs={}
    for halo in (0,1):
        lib.gcc_set_option(_lib.OPT_IGEMM_HALO, halo)
        y = ops.new_act(N, Co, Ho, Wo, DEV); y.fill_(7.0)
        dx = ops.new_act(N, Ci, H, W, DEV); dx.fill_(7.0)
        ops.conv_fprop(x, w, Co, k, s, p, out=y)
        ops.conv_dgrad(dy, wt, Ci, H, W, k, s, p, out=dx)
        torch.cuda.synchronize()
        outs[halo]=(y.float().clone(), dx.float().clone())
    ref_y = torch.nn.functional.conv2d(x.float(), m.bfloat16().float(), stride=2, padding=1)
    ref_dx = torch.nn.grad.conv2d_input((N,Ci,H,W), m.bfloat16().float(), dy.float(), stride=2, padding=1)
    for name,i,ref in (('fprop',0,ref_y),('dgrad',1,ref_dx)):
        a,b=outs[0][i],outs[1][i]
        sc=ref.abs().max().item()
        print(f'{N}x{Ci}->{Co}@{H} {name}: igemm-vs-ref {((a-ref).abs().max()/sc).item():.2e} halo-vs-ref {((b-ref).abs().max()/sc).item():.2e} halo-vs-igemm {((a-b).abs().max()/sc).item():.2e}  mismatching elems {(a!=b).float().mean().item():.4f}')
run(16,128,128,128,256)
run(16,64,64,256,512)
run(2,64,64,128,256)
run(4,32,32,64,256)
