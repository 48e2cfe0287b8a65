"""GPU probe: run the config-3 training step until a loss or gradient goes non-finite; report the iteration and the tensors involved."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene

sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
g = sc.geometry
named = dict(g.named_parameters())
named['tex.table'] = sc.material['kd_ks'].encoder.params
for k, p in sc.material['kd_ks'].net.named_parameters():
    named['tex.' + k] = p
named['trans'] = sc.FLAGS.trans_optim
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 700):
    r = sc.step()
    bad = [k for k, v in r.items() if not torch.isfinite(v).all()]
    gbad = [(k, float(p.grad.abs().max())) for k, p in named.items() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gm = {k: float(p.grad.abs().max()) for k, p in named.items() if p.grad is not None and torch.isfinite(p.grad).all()}
    gmax = max(gm.values(), default=0.0)
    top = sorted(gm.items(), key=lambda kv: -kv[1])[:3]
    pbad = [k for k, p in named.items() if not torch.isfinite(p).all()]
    tops = ' '.join('%s=%.1e' % kv for kv in top)
    if it % 50 == 0 or bad or gbad or pbad or gmax > 1e6:
        md = g.last_mesh_dict
        print(f'it {it} total {float(r["total"]):.4f} msk {float(r["msk_loss"]):.3f} verts {md["imesh"].v_pos.shape[0]} gmax {gmax:.3e} top {tops} bad_loss {bad} bad_grad {gbad[:4]} bad_param {pbad[:4]}', flush=True)
    if bad or gbad or pbad:
        break
