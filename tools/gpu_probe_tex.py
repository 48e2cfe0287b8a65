"""dev tool: texture-MLP kernel timings on a realistic pixel distribution (4 x 1024^2, ~15% covered, surface-like positions)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import texmlp
dev='cuda'; torch.manual_seed(0)
B,H,W=4,1024,1024
ys,xs=torch.meshgrid(torch.linspace(-1.3,0.45,H,device=dev),torch.linspace(-0.9,0.9,W,device=dev),indexing='ij')
pos=torch.stack([xs,ys,0.1*torch.sin(3*xs)*torch.cos(2*ys)],-1)[None].expand(B,-1,-1,-1).contiguous()
mask=((xs.abs()<0.25+0.1*torch.cos(4*ys))&(ys<0.4)&(ys>-1.25)).float()[None,...,None].expand(B,-1,-1,-1).contiguous()
print('coverage',mask.mean().item())
npar=texmlp.grid_param_count()
bbox=(0.6,0.6,0.2,-0.8,-1.2,-0.2); omin,omax=(0,0,0,0,0.001,0),(1,1,1,0,1,1)
def T(fn,K=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/K
for gt,gw,gx in [(False,False,False),(True,False,False),(False,True,False),(False,False,True),(True,True,True)]:
    table=((torch.rand(npar,device=dev)*2-1)*1e-2).requires_grad_(gt)
    w1=(torch.randn(32,10,device=dev)*0.5).requires_grad_(gw); w2=(torch.randn(32,32,device=dev)*0.3).requires_grad_(gw); w3=(torch.randn(6,32,device=dev)*0.3).requires_grad_(gw)
    p=pos.clone().requires_grad_(gx)
    if not (gt or gw or gx):
        print('fwd only', T(lambda: texmlp.texture_mlp(p,table,w1,w2,w3,bbox,omin,omax,mask=mask))); continue
    def fb():
        o=texmlp.texture_mlp(p,table,w1,w2,w3,bbox,omin,omax,mask=mask); o.sum().backward()
    print(f'fwd+bwd table={gt} w={gw} x={gx}:', T(fb))
