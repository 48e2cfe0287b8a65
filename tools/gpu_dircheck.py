"""GPU probe: directional-derivative check of the end-to-end gradients of one init-stage iteration (reduced size): for every parameter
group the loss change along -g/|g| is compared with the first-order prediction -eps |g|.  Coverage is discrete and several kernels
accumulate with unordered atomics, so agreement is to ~10-20 % at eps 1e-4 .. 1e-3 and the loss itself has ~1e-4 run-to-run noise."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'d3human-code_amd'))
import torch
from d3h import scene
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=False, res=512, grid_n=40, n_frames=2, loss_set='full', body_verts=4096)
g=sc.geometry
bg=torch.rand(2,512,512,3,device='cuda')
def loss():
    torch.manual_seed(0)
    tgt=sc.target(bg)
    r=g.tick_init(sc.glctx,tgt,None,sc.material,sc.loss_fn,5,None)
    return r, r['reg_loss']+r['normal_loss']+r['msk_loss']+r['ssim_loss']
groups={'sdf_net':list(g.sdf_net.parameters()),'deform':[g.deform],'msdf':[g.msdf],'tex':list(sc.material['kd_ks'].parameters()),'trans':[sc.FLAGS.trans_optim]}
allp=[p for ps in groups.values() for p in ps]
for p in allp: p.grad=None
r,L=loss(); L.backward()
print('L0',float(L),{k:round(float(v),5) for k,v in r.items()})
for name,ps in groups.items():
    gs=[p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in ps]
    gn=torch.sqrt(sum((x**2).sum() for x in gs))
    if gn==0: print(name,'zero grad'); continue
    for eps in (1e-3,1e-4):
        with torch.no_grad():
            for p,x in zip(ps,gs): p.add_(x/gn*(-eps))
            _,L1=loss()
            for p,x in zip(ps,gs): p.add_(x/gn*(eps))
        print(name,'|g|',float(gn),'eps',eps,'predicted dL',float(-eps*gn),'actual dL',float(L1-L))
