"""per-term / per-stage comparison of the product tick with the oracle chain (debug aid for tests/test_gpu_e2e.py)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import e2e_cases as E
from oracle import tick as OTK
dev = sys.argv[1] if len(sys.argv) > 1 else 'cuda'
if dev == 'cpu':
    from d3h import _lib as L
    L._use_emulator_for_tests(os.path.join(ROOT, 'tests/emul/libd3h_emul.so'))
kw = dict(n=14, res=80, frames=2, seed=0)
if len(sys.argv) > 2:
    kw.update(eval(sys.argv[2]))
st = E.make_state(**kw)
buffers = ('shaded', 'geometric_normal', 'msdf_image')
ro = OTK.tick_init(st, buffers=buffers, keep=True)
P = E.build_product(dev, st, 2 * kw['n'], buffers)
r, total = E.product_tick(P, st, dev)
F, g = P['FLAGS'], P['geometry']
d = g.last_mesh_dict
posed_p = d['deform_imesh'].v_pos
print('posed max diff', float((posed_p.detach().cpu() - ro['_mesh']['posed'].detach()).abs().max()))
for k in ('msk_loss', 'img_loss', 'normal_loss', 'ssim_loss', 'sdf_reg_loss', 'eik_loss'):
    gp, = torch.autograd.grad(r[k], F.trans_optim, retain_graph=True, allow_unused=True)
    go, = torch.autograd.grad(ro[k], st['trans'], retain_graph=True, allow_unused=True)
    print(k, 'value', float(r[k]), float(ro[k]))
    if gp is not None and go is not None:
        print('   d/dtrans product', gp.cpu().reshape(-1).tolist())
        print('   d/dtrans oracle ', go.reshape(-1).tolist())
    gpp, = torch.autograd.grad(r[k], posed_p, retain_graph=True, allow_unused=True)
    gop, = torch.autograd.grad(ro[k], ro['_mesh']['posed'], retain_graph=True, allow_unused=True)
    if gpp is not None and gop is not None:
        e = (gpp.cpu() - gop).abs()
        i = int(e.reshape(-1).argmax())
        print('   d/dposed max|oracle| %.3e  max err %.3e at flat %d; n(err>1e-3 max) %d' % (float(gop.abs().max()), float(e.max()), i, int((e > 1e-3 * gop.abs().max()).sum())))
        fr, v = i // (3 * gop.shape[1]), (i // 3) % gop.shape[1]
        print('     worst vertex frame %d id %d product %s oracle %s' % (fr, v, gpp[fr, v].tolist(), gop[fr, v].tolist()))
