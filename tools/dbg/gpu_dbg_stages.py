"""stage-by-stage gradient comparison (product vs oracle chain) of one loss term"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import e2e_cases as E
from oracle import tick as OTK
dev = sys.argv[1] if len(sys.argv) > 1 else 'cuda'
term = sys.argv[2] if len(sys.argv) > 2 else 'ssim_loss'
if dev == 'cpu':
    from d3h import _lib as L
    L._use_emulator_for_tests(os.path.join(ROOT, 'tests/emul/libd3h_emul.so'))
kw = dict(n=14, res=80, frames=2, seed=0)
if len(sys.argv) > 3:
    kw.update(eval(sys.argv[3]))
st = E.make_state(**kw)
buffers = ('shaded', 'geometric_normal', 'msdf_image')
import render.render as RR
import d3h.raster as R_, d3h.imgops as I_
cap = {}
orig_aa, orig_gb, orig_xfm = RR.dr.antialias, RR._R.gbuffer, RR.ru.xfm_points
def aa(color, rast, pos, tri):
    cap['pre_aa'], cap['clip_in_aa'] = color, pos
    o = orig_aa(color, rast, pos, tri); cap['post_aa'] = o; return o
def gb(*a, **k):
    o = orig_gb(*a, **k); cap['gb'] = o; return o
def xfm(p, m):
    o = orig_xfm(p, m); cap['clip'] = o; return o
RR.dr.antialias, RR._R.gbuffer, RR.ru.xfm_points = aa, gb, xfm
P = E.build_product(dev, st, 2 * kw['n'], buffers + ('_rast',))
r, total = E.product_tick(P, st, dev)
rast_p = P['geometry'].last_mesh_dict['buffers']['_rast'].detach().cpu()
ro = OTK.tick_init(st, buffers=buffers, keep=True, rast_zw=rast_p[..., 2])
S = ro['_stages']
print('id diffs', torch.nonzero(rast_p[..., 3] != S['rast'][..., 3]).tolist(), 'relu kinks', E.relu_kinks(st, ro))
def cmp(name, a, b):
    a, b = a.detach().cpu(), b.detach()
    e = (a - b).abs()
    print('%-28s max|oracle| %.3e max err %.3e (rel %.1e) n(err > 1e-3 max) %d of %d' % (name, float(b.abs().max()), float(e.max()), float(e.max() / b.abs().max()), int((e > 1e-3 * b.abs().max()).sum()), e.numel()))
    return e
print('keys', S['keys'])
cmp('value post_aa', cap['post_aa'], S['post_aa'])
cmp('value pre_aa', cap['pre_aa'], S['pre_aa'])
cmp('value clip', cap['clip'], S['clip'])
G = lambda y, x: torch.autograd.grad(y, x, retain_graph=True, allow_unused=True)[0]
e = cmp('d/d post_aa', G(r[term], cap['post_aa']), G(ro[term], S['post_aa']))
e = cmp('d/d pre_aa', G(r[term], cap['pre_aa']), G(ro[term], S['pre_aa']))
idx = torch.nonzero(e > 1e-3 * float(G(ro[term], S['pre_aa']).abs().max()))
print('   worst pre_aa entries (b, y, x, c):', idx[:12].tolist())
e = cmp('d/d clip', G(r[term], cap['clip']), G(ro[term], S['clip']))
idx = torch.nonzero(e > 1e-3 * float(G(ro[term], S['clip']).abs().max()))
print('   worst clip entries (b, v, c):', idx[:12].tolist())
groups = cap['gb'][0]
gpo = [g_ for g_ in groups if g_.shape[-1] == 3][0]
e = cmp('d/d gb_pos_orig', G(r[term], gpo), G(ro[term], S['gb_pos_orig']))
e = (cap['post_aa'].detach().cpu() - S['post_aa'].detach()).abs()
idx = torch.nonzero(e > 1e-4)
print('post_aa value mismatches:', idx.tolist())
rast = S['rast']
seen = set()
for b, y, x, c in idx.tolist():
    if (b, y, x) in seen:
        continue
    seen.add((b, y, x))
    print(' pixel', (b, y, x), 'product', cap['post_aa'][b, y, x].tolist(), '\n   oracle', S['post_aa'][b, y, x].tolist(), '\n   pre', S['pre_aa'][b, y, x].tolist())
    for dy, dx in ((0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)):
        yy, xx = y + dy, x + dx
        print('   nb', (dy, dx), 'rast', rast[b, yy, xx].tolist(), 'pre', [round(v, 4) for v in S['pre_aa'][b, yy, xx].tolist()])
    tid = int(rast[b, y, x, 3])
    faces = ro['_mesh']['faces']
    clip = S['clip'].detach()
    for t in {int(rast[b, y + dy, x + dx, 3]) for dy, dx in ((0, 0), (0, 1), (0, -1), (1, 0), (-1, 0))}:
        if t > 0:
            vid = faces[t - 1].tolist()
            P_ = clip[b, vid]
            sx = (P_[:, 0] / P_[:, 3] * 0.5 + 0.5) * 80
            sy = (P_[:, 1] / P_[:, 3] * 0.5 + 0.5) * 80
            print('   tri', t, 'verts', vid, 'sx', sx.tolist(), 'sy', sy.tolist())
from oracle import texmlp as OT
go = G(ro[term], S['gb_pos_orig']); gp = G(r[term], gpo).cpu()
e = (gp - go).abs().max(-1).values
idx = torch.nonzero(e > 1e-3 * float(go.abs().max()))
lay, _ = OT.grid_layout()
mat = st['material']
b0, b1 = torch.tensor(mat['bbox'][:3]), torch.tensor(mat['bbox'][3:])
for b, y, x in idx.tolist():
    for name, src in (('oracle', S['gb_pos_orig'].detach()), ('product', gpo.detach().cpu())):
        xn = torch.clamp((src[b, y, x] - b0) / (b1 - b0), 0, 1)
        print(name, (b, y, x), 'pos', src[b, y, x].tolist(), 'frac per level', [['%.6f' % v for v in ((xn * sc + 0.5) - torch.floor(xn * sc + 0.5)).tolist()] for sc, _, _, _ in lay])
    print('   grad product', gp[b, y, x].tolist(), 'oracle', go[b, y, x].tolist())
