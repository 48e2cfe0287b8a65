import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, numpy as np
import e2e_cases as E
from oracle import tick as OTK, raster as OR, render as ORD, texmlp as OT
from d3h import texmlp
st = E.make_state(n=14, res=80, frames=2, seed=0)
with torch.no_grad():
    m = OTK.get_mesh_init(st, [0, 1])
    clip = ORD.xfm_points(m['posed'], st['mvp'])
    rast, _ = OR.rasterize(clip, m['faces'], 80, 80)
    gpo, _ = OR.interpolate(m['verts'][None], rast, m['faces'])
cov = rast[..., 3] > 0
x = gpo[cov]                                     # covered pixels only
mat = st['material']
gen = torch.Generator().manual_seed(0)
G = torch.randn(x.shape[0], 6, generator=gen)
xr = x.clone().requires_grad_(True)
(OT.texture_mlp(xr, mat['table'].detach(), mat['w1'].detach(), mat['w2'].detach(), mat['w3'].detach(), mat['bbox'], mat['omin'], mat['omax']) * G).sum().backward()
xa = x.clone().cuda().requires_grad_(True)
(texmlp.texture_mlp(xa, mat['table'].detach().cuda(), mat['w1'].detach().cuda(), mat['w2'].detach().cuda(), mat['w3'].detach().cuda(), mat['bbox'], mat['omin'], mat['omax']) * G.cuda()).sum().backward()
e = (xa.grad.cpu() - xr.grad).abs().max(1).values
den = xr.grad.abs().max()
bad = torch.nonzero(e > 1e-3 * den).reshape(-1)
print('pixels', x.shape[0], 'bad', bad.numel(), 'max rel', float(e.max() / den))
lay, _ = OT.grid_layout()
b0, b1 = torch.tensor(mat['bbox'][:3]), torch.tensor(mat['bbox'][3:])
for i in bad[:8].tolist():
    xn = torch.clamp((x[i] - b0) / (b1 - b0), 0, 1)
    fr = [((xn * sc + 0.5) - torch.floor(xn * sc + 0.5)).tolist() for sc, _, _, _ in lay]
    print(i, 'dx', xa.grad[i].tolist(), 'ref', xr.grad[i].tolist())
    print('    fractional parts per level:', [['%.6f' % v for v in f] for f in fr])
