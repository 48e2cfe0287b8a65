"""GPU probe: which loss term carries the 1e15 gradient spikes (d term / d trans per term, every iteration)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene

sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=False)
F = sc.FLAGS
for it in range(40):
    bg = torch.rand(sc.n_frames, sc.res, sc.res, 3, device=sc.device)
    tgt = sc.target(bg)
    sc._zero_grad()
    r = sc.geometry.tick_init(sc.glctx, tgt, None, sc.material, sc.loss_fn, sc.it, None)
    out = []
    for k in ('msk_loss', 'normal_loss', 'ssim_loss', 'img_loss', 'reg_loss'):
        g, = torch.autograd.grad(r[k], F.trans_optim, retain_graph=True, allow_unused=True)
        out.append('%s=%.1e' % (k, float(g.abs().max()) if g is not None else 0.0))
    total = r['d3h_total'] if 'd3h_total' in r else r['reg_loss'] + r['normal_loss'] + r['msk_loss'] + r.get('ssim_loss', 0.0)
    total.backward()
    sc._optimizer_step()
    sc.it += 1
    print(it, ' '.join(out), flush=True)
