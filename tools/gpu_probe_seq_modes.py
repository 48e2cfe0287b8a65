import sys, os
sys.path.insert(0, '/root/repo/d3human-code_amd'); sys.path.insert(0, '/root/repo')
import torch
from d3h.scene import Scene
sc = Scene(res=512, grid_n=32, n_frames=1, device='cuda', prefit_steps=0, loss_set='seq', body_verts=4096)
for i in range(4):
    r = sc.step_seq()
g = sc.geometry
bg = torch.rand(1, 512, 512, 3, device='cuda')
def terms():
    torch.manual_seed(0)
    tgt = sc.target(bg)
    tgt.update({'cloth_img': sc.cloth_img, 'body_img': sc.body_img})
    return g.tick_seq(sc.glctx, tgt, None, sc.material, sc.loss_fn, 5, None, t='all')
for rep in range(3):
    a = terms()
    with torch.no_grad():
        b = terms()
    for k in ('nds_normal_loss', 'delta_loss', 'laplacian_loss', 'colli_loss'):
        print(rep, k, float(a[k]), float(b[k]), float(a[k]) - float(b[k]))
