"""full-size whole-tick parity, broken down by loss term: d(term)/d(deform, SDF net) of the GPU tick vs the oracle tick (shared raster
decisions) at 1 frame x 1024^2, tet-res 128 -- which term carries the per-cent-level gradient differences?
    python tools/dbg/gpu_dbg_parity_terms.py [grid_n res]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from d3h import scene
from oracle import parity as OP, tick as OTK, render as ORD
n, res = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (63, 1024)
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=res, grid_n=n, n_frames=1, loss_set='full')
for _ in range(5):
    sc.step()
if os.environ.get('TABLE_AMP'):
    sc.material['kd_ks'].encoder.params.data.uniform_(-float(os.environ['TABLE_AMP']), float(os.environ['TABLE_AMP']))
g = sc.geometry
gen = torch.Generator().manual_seed(1001)
bg = torch.rand(1, res, res, 3, generator=gen)
torch.manual_seed(2001)
draws = ORD.draw_jitter(1, res, res)
base = ('shaded', 'geometric_normal', 'msdf_image')
sc.FLAGS.render_buffers = base + ('_rast',)
store = []


def gpu_tick(pts=None):
    sc._zero_grad()
    sc.FLAGS.trans_optim.grad = None
    ctx = OP.recorded_surface_samples(store) if pts is None else OP.fixed_surface_samples(pts)
    with ctx, OP.fixed_render_draws([draws], 'cuda'):
        return g.tick_init(sc.glctx, sc.target(bg.cuda()), None, sc.material, sc.loss_fn, 10, None)


r = gpu_tick()
pts = store[0]
rast_p = g.last_mesh_dict['buffers']['_rast'].detach().cpu()
st = OP.state_from_scene(sc, bg, pts, 10)
ro = OTK.tick_init(st, buffers=base, draws=draws, keep=True, rast_zw=rast_p[..., 2], rast_ids=rast_p[..., 3])
import e2e_cases as E
print('relu kinks (|pre-activation| < 4e-6):', E.relu_kinks(st, ro), ' covered pixels', int((rast_p[..., 3] > 0).sum()))
names = ['deform'] + ['sd.' + k for k, _ in g.sdf_net.state_dict().items()] + ['trans']
params_p = [g.deform] + list(g.sdf_net.parameters()) + [sc.FLAGS.trans_optim]
params_o = [st['deform']] + [st['sd'][k] for k, _ in g.sdf_net.state_dict().items()] + [st['trans']]
for term in ('msk_loss', 'img_loss', 'normal_loss', 'ssim_loss', 'eik_loss', 'sdf_reg_loss'):
    r = gpu_tick(pts)
    r[term].backward()
    gp = [p.grad for p in params_p]
    go = torch.autograd.grad(ro[term], params_o, retain_graph=True, allow_unused=True)
    out = {}
    for name, a, b in zip(names, gp, go):
        if name not in ('deform', 'sd.net.0.weight', 'sd.net.14.weight', 'trans'):
            continue
        if a is None or b is None:
            out[name] = None
            continue
        a, b = a.detach().cpu().double(), b.double()
        if torch.isnan(a).any() or torch.isnan(b).any():
            out[name] = 'nan: gpu %d oracle %d of %d' % (int(torch.isnan(a).sum()), int(torch.isnan(b).sum()), a.numel())
            continue
        out[name] = (float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)), float((a - b).norm() / max(float(b.norm()), 1e-30)))
    print(term, 'gpu %.8f oracle %.8f' % (float(r[term]), float(ro[term])), json.dumps(out))
