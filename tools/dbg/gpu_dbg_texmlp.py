import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT)
import torch
from d3h import texmlp
from oracle import texmlp as OT
gen = torch.Generator().manual_seed(77)
npar = texmlp.grid_param_count()
table = (torch.rand(npar, generator=gen) * 2 - 1) * 0.5
w = [torch.randn(32, 10, generator=gen) * 0.5, torch.randn(32, 32, generator=gen) * 0.3, torch.randn(6, 32, generator=gen) * 0.3]
bbox = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)
omin, omax = (0, 0, 0, 0, 0.001, 0), (1, 1, 1, 0, 1, 1)
lay, _ = OT.grid_layout()
for n in (5000, 70000, 300000):
    x = torch.rand(n, 3, generator=gen) * torch.tensor([1.4, 1.8, 0.4]) + torch.tensor([-0.8, -1.2, -0.2])
    G = torch.randn(n, 6, generator=gen)
    rt = table.clone().requires_grad_(True); rx = x.clone().requires_grad_(True)
    (OT.texture_mlp(rx, rt, w[0], w[1], w[2], bbox, omin, omax) * G).sum().backward()
    rt64 = table.double().clone().requires_grad_(True)
    for mode in ('async', 'sync'):
        texmlp.ASYNC_TABLE_GRAD = (mode == 'async')
        tab = table.clone().cuda().requires_grad_(True); xa = x.clone().cuda().requires_grad_(True)
        ws = [t.clone().cuda().requires_grad_(True) for t in w]
        (texmlp.texture_mlp(xa, tab, ws[0], ws[1], ws[2], bbox, omin, omax) * G.cuda()).sum().backward()
        torch.cuda.synchronize()
        d = (tab.grad.cpu() - rt.grad)
        per = []
        for (scale, res, off, size) in lay:
            sl = slice(2 * off, 2 * (off + size))
            per.append('%.1e/%.1e' % (float(d[sl].abs().max()), float(rt.grad[sl].abs().max())))
        print(n, mode, 'table err/max per level', per, 'dx rel %.1e' % float((xa.grad.cpu() - rx.grad).abs().max() / rx.grad.abs().max()), flush=True)
