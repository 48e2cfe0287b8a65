"""where do the per-cent-level differences of the SDF weight / bias gradients between the GPU tick and the oracle tick
(shared raster) come from?  On REPRODUCIBLE states (tests/golden/parity_state_sdf.npz + Scene.perturb_state_seeded) at the config-3 shape
(1 frame, tet-res 128, 1024^2, full loss set), per loss term:

  * d(term)/d(sdf) over the grid from both sides (GPU: the `gout` the sweep's backward receives; oracle: autograd.grad): its plain sum IS the
    head-bias gradient -- difference of the sums, L1 of the difference, the largest entries, how many entries carry the difference;
  * d(term)/d(buffer pixel) AFTER antialias from both sides: which pixels carry the difference, with the length of the pre-normalisation
    normal and the alpha there.

    python tools/dbg/gpu_parity_rootcause.py [seeds, comma separated] [grid_n res]
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch                                               # noqa: E402
from d3h import scene                                      # noqa: E402
from oracle import parity as OP, tick as OTK, render as ORD  # noqa: E402

DEV = 'cuda'
if os.environ.get('D3H_EMUL') == '1':          # toy sizes on the host emulation of the kernels: debugging this script without a GPU
    from d3h import _lib as L
    L._use_emulator_for_tests(os.path.join(ROOT, 'tests', 'emul', 'libd3h_emul.so'))
    DEV = 'cpu'
seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else '0').split(',')]
n, res = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (63, 1024)
STATE = os.path.join(ROOT, 'tests', 'golden', 'parity_state_sdf.npz')
TERMS = ('msk_loss', 'normal_loss', 'ssim_loss', 'sdf_reg_loss', 'eik_loss')
base = ('shaded', 'geometric_normal', 'msdf_image')


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)), float((a - b).norm() / max(float(b.norm()), 1e-30))


for seed in seeds:
    sc = scene.Scene(device=DEV, visualize_watertight=True, res=res, grid_n=n, n_frames=1, loss_set='full', sdf_state=STATE,
                     flags_hook=(lambda F: setattr(F, 'eikonal_samples', int(os.environ['D3H_EIK_SAMPLES']))) if os.environ.get('D3H_EIK_SAMPLES') else None)
    sc.perturb_state_seeded(seed)
    sc.set_kinkfree_texture(seed)
    g = sc.geometry
    gen = torch.Generator().manual_seed(1000 + seed)
    bg = torch.rand(1, res, res, 3, generator=gen)
    torch.manual_seed(2000 + seed)
    draws = ORD.draw_jitter(1, res, res)
    sc.FLAGS.render_buffers = base + ('_rast',)
    store = []
    cap = {}

    def gpu_tick(pts=None):
        sc._zero_grad()
        sc.FLAGS.trans_optim.grad = None
        ctx = OP.recorded_surface_samples(store) if pts is None else OP.fixed_surface_samples(pts)
        with ctx, OP.fixed_render_draws([draws], DEV):
            r = g.tick_init(sc.glctx, sc.target(bg.to(DEV)), None, sc.material, sc.loss_fn, 10, None)
        d = g.last_mesh_dict
        cap.clear()
        d['sdf'].register_hook(lambda t: cap.__setitem__('sdf', t.detach().clone()))
        for nm_, t_ in (('posed', d['deform_imesh'].v_pos), ('verts', d['imesh'].v_pos)):
            if t_.requires_grad:
                t_.register_hook(lambda t, nm_=nm_: cap.__setitem__(nm_, t.detach().clone()))
        st_ = d['buffers']['_stacked']
        if st_ is not None and st_.requires_grad:
            st_.register_hook(lambda t: cap.__setitem__('stacked', t.detach().clone()))
        return r, d

    r, d = gpu_tick()
    pts = store[0]
    layout = dict(d['buffers']['_layout'])
    rast_p = d['buffers']['_rast'].detach().cpu()
    faces_p = d['imesh'].t_pos_idx.detach().cpu().long()
    st = OP.state_from_scene(sc, bg, pts, 10)
    ro = OTK.tick_init(st, buffers=base, draws=draws, keep=True, rast_zw=rast_p[..., 2], rast_ids=rast_p[..., 3])
    # the same oracle tick in float64: a reference for the reference
    st64 = OP._to_dtype(OP.state_from_scene(sc, bg, pts, 10), torch.float64)
    torch.set_default_dtype(torch.float64)
    try:
        ro64 = OTK.tick_init(st64, buffers=base, draws=OP._to_dtype(draws, torch.float64), keep=True, rast_zw=rast_p[..., 2].double(), rast_ids=rast_p[..., 3].double())
    finally:
        torch.set_default_dtype(torch.float32)
    print('float64 oracle: same mesh', bool(torch.equal(ro64['_mesh']['faces'], ro['_mesh']['faces'])), flush=True)
    faces_equal = bool(faces_p.shape == ro['_mesh']['faces'].shape and torch.equal(faces_p, ro['_mesh']['faces']))
    a_p = d['buffers']['shaded'][..., 3].detach().cpu()
    a_o = ro['_buffers']['shaded'][..., 3].detach()
    alpha_bad = torch.nonzero((a_p - a_o).abs() > 1e-3)
    gn_o = ro['_buffers']['geometric_normal'][..., 0:3].detach()
    gn_p = d['buffers']['geometric_normal'][..., 0:3].detach().cpu()
    len_o = gn_o.norm(dim=-1)
    thin = (len_o > 0) & (len_o < 0.05)
    print(f'STATE seed {seed}: faces {faces_p.shape[0]} equal {faces_equal}; covered {int((rast_p[..., 3] > 0).sum())}; alpha kinks (shared) {alpha_bad.shape[0]}; '
          f'pixels with 0 < |geometric_normal| < 0.05: {int(thin.sum())} (< 0.01: {int(((len_o > 0) & (len_o < 0.01)).sum())}, < 1e-3: {int(((len_o > 0) & (len_o < 1e-3)).sum())}); '
          f'max |gn_gpu - gn_oracle| {float((gn_p - gn_o).abs().max()):.3e}', flush=True)
    names = ['deform', 'msdf'] + ['sd.' + k for k, _ in g.sdf_net.state_dict().items()] + ['trans']
    params_p = [g.deform, g.msdf] + list(g.sdf_net.parameters()) + [sc.FLAGS.trans_optim]
    params_o = [st['deform'], st['msdf']] + [st['sd'][k] for k, _ in g.sdf_net.state_dict().items()] + [st['trans']]
    params_o64 = [st64['deform'], st64['msdf']] + [st64['sd'][k] for k, _ in g.sdf_net.state_dict().items()] + [st64['trans']]
    sdf_o64 = ro64['_mesh']['sdf']
    tot_o64 = None
    sdf_o = ro['_mesh']['sdf']
    bufs_o = [ro['_buffers'][k] for k in base]
    tot_p, tot_o = None, None
    for term in TERMS:
        r, d = gpu_tick(pts)
        r[term].backward()
        if DEV == 'cuda':
            torch.cuda.synchronize()
        gp = [None if p.grad is None else p.grad.detach().cpu().clone() for p in params_p]
        go = torch.autograd.grad(ro[term], params_o + [sdf_o] + bufs_o, retain_graph=True, allow_unused=True)
        go_par, go_sdf, go_buf = go[:len(params_o)], go[len(params_o)], go[len(params_o) + 1:]
        go64 = torch.autograd.grad(ro64[term], params_o64 + [sdf_o64, ro64['_mesh']['posed'], ro64['_mesh']['verts']], retain_graph=True, allow_unused=True)
        go64_par, go64_sdf = go64[:len(params_o64)], go64[len(params_o64)]
        go32_mesh = torch.autograd.grad(ro[term], [ro['_mesh']['posed'], ro['_mesh']['verts']], retain_graph=True, allow_unused=True)
        if 'verts' in cap and 'sdf' in cap and go64_sdf is not None and term in ('msk_loss', 'normal_loss', 'ssim_loss'):
            # isolate the marching-tets backward: the GPU's own d/d(verts) pushed through the float64 oracle's extraction, against the GPU's d/d(sdf)
            gv = cap['verts'].cpu().double().reshape(ro64['_mesh']['verts'].shape)
            via64, = torch.autograd.grad(ro64['_mesh']['verts'], sdf_o64, grad_outputs=gv, retain_graph=True)
            a_, b_ = cap['sdf'].cpu().double().reshape(-1), via64.reshape(-1)
            print('  marching-tets backward alone (GPU d/dverts -> float64 extraction backward vs GPU d/dsdf): sums %.7e vs %.7e ; L1 diff %.3e of %.3e ; signed parts +%.3e / %.3e' % (
                float(a_.sum()), float(b_.sum()), float((a_ - b_).abs().sum()), float(b_.abs().sum()), float((a_ - b_)[(a_ - b_) > 0].sum()), float((a_ - b_)[(a_ - b_) < 0].sum())), flush=True)
            gp_ = cap['posed'].cpu().double().reshape(ro64['_mesh']['posed'].shape)
            via64v, = torch.autograd.grad(ro64['_mesh']['posed'], ro64['_mesh']['verts'], grad_outputs=gp_, retain_graph=True)
            c64_ = go64[-2]
            if c64_ is not None:
                e_ = (gp_ - c64_).reshape(-1, 3)
                g_ = c64_.reshape(-1, 3)
                gn_ = g_.norm(dim=1).clamp(min=1e-30)
                along = (e_ * g_).sum(1) / gn_                      # error component along the true gradient
                print('  d/dposed error: sum of the component ALONG the float64 gradient %.4e (|.| %.4e) vs sum |g| %.4e -> relative magnitude bias %.3e ; perpendicular L1 %.3e ; nonzero rows %d' % (
                    float(along.sum()), float(along.abs().sum()), float(gn_.sum()), float(along.sum() / gn_.sum()), float((e_ - along[:, None] * g_ / gn_[:, None]).norm(dim=1).sum()),
                    int((g_.norm(dim=1) > 0).sum())), flush=True)
                top_ = torch.topk(e_.norm(dim=1), 5).indices
                for i_ in top_.tolist():
                    print('      vert %d gpu %s f64 %s' % (i_, ['%.6e' % v for v in gp_.reshape(-1, 3)[i_].tolist()], ['%.6e' % v for v in g_[i_].tolist()]))
            print('  LBS backward alone (GPU d/dposed -> float64 LBS backward vs GPU d/dverts): %s' % ['%.1e' % v for v in rel(gv, via64v)], flush=True)
        for nm_, c64, c32 in (('posed', go64[-2], go32_mesh[0]), ('verts', go64[-1], go32_mesh[1])):
            if c64 is not None and c32 is not None and nm_ in cap:
                a_ = cap[nm_].cpu().double().reshape(c64.shape)
                print('  dT/d%s: gpu vs f64 (max, l2) %s ; oracle32 vs f64 %s ; gpu vs oracle32 %s ; sums gpu %s f64 %s' % (
                    nm_, ['%.1e' % v for v in rel(a_, c64)], ['%.1e' % v for v in rel(c32, c64)], ['%.1e' % v for v in rel(a_, c32)],
                    ['%.6e' % float(v) for v in a_.reshape(-1, 3).sum(0)], ['%.6e' % float(v) for v in c64.reshape(-1, 3).sum(0)]), flush=True)
        line = {'term': term, 'gpu': float(r[term]), 'oracle': float(ro[term]), 'oracle64': float(ro64[term])}
        for name, a, b, c in zip(names, gp, go_par, go64_par):
            if name in ('deform', 'msdf', 'sd.net.0.weight', 'sd.net.8.weight', 'sd.net.14.weight', 'sd.net.14.bias', 'sd.net.0.bias', 'trans') and a is not None and b is not None:
                # [gpu vs oracle32 (max, l2)], [gpu vs float64], [oracle32 vs float64], max |.|
                line[name] = ['%.1e' % v for v in rel(a, b)] + ['|'] + ['%.1e' % v for v in rel(a, c)] + ['|'] + ['%.1e' % v for v in rel(b, c)] + ['%.2e' % float(b.abs().max())]
        print('TERM', json.dumps(line), flush=True)
        if tot_o64 is None:
            tot_o64 = [None if c is None else c.clone() for c in go64_par]
        else:
            for i, c in enumerate(go64_par):
                if c is not None:
                    tot_o64[i] = c.clone() if tot_o64[i] is None else tot_o64[i] + c
        if go64_sdf is not None and go_sdf is not None and 'sdf' in cap:
            a, b, c = cap['sdf'].cpu().double().reshape(-1), go_sdf.double().reshape(-1), go64_sdf.reshape(-1)
            print('  dT/dsdf sums: gpu %.7e oracle32 %.7e float64 %.7e | L1 of (gpu - f64) %.3e, (oracle32 - f64) %.3e, (gpu - oracle32) %.3e of L1 %.3e' % (
                float(a.sum()), float(b.sum()), float(c.sum()), float((a - c).abs().sum()), float((b - c).abs().sum()), float((a - b).abs().sum()), float(c.abs().sum())), flush=True)
        if tot_p is None:
            tot_p = [None if a is None else a.clone().double() for a in gp]
            tot_o = [None if b is None else b.clone().double() for b in go_par]
        else:
            for i, (a, b) in enumerate(zip(gp, go_par)):
                if a is not None:
                    tot_p[i] = a.double() if tot_p[i] is None else tot_p[i] + a.double()
                if b is not None:
                    tot_o[i] = b.double() if tot_o[i] is None else tot_o[i] + b.double()
        # d(term)/d(sdf) over the grid
        if go_sdf is not None and 'sdf' in cap:
            a, b = cap['sdf'].cpu().double().reshape(-1), go_sdf.double().reshape(-1)
            e = a - b
            top = torch.topk(e.abs(), 8).indices
            print('  dT/dsdf: nonzero gpu %d oracle %d; sum gpu %.6e oracle %.6e (diff %.3e = %.2e of sum|.|); L1 diff %.3e of L1 %.3e; max|.| %.3e; entries above 1e-3 of max: %d, above 1e-4: %d' % (
                int((a != 0).sum()), int((b != 0).sum()), float(a.sum()), float(b.sum()), float(e.sum()), float(e.sum().abs() / max(float(b.abs().sum()), 1e-30)),
                float(e.abs().sum()), float(b.abs().sum()), float(b.abs().max()), int((e.abs() > 1e-3 * b.abs().max()).sum()), int((e.abs() > 1e-4 * b.abs().max()).sum())))
            print('     top entries (idx, gpu, oracle):', [(int(i), '%.4e' % float(a[i]), '%.4e' % float(b[i])) for i in top.tolist()])
            pos = e[e > 0].sum(); neg = e[e < 0].sum()
            print('     signed parts of the difference: +%.3e / %.3e; top-8 carry %.3e of L1 %.3e' % (float(pos), float(neg), float(e.abs()[top].sum()), float(e.abs().sum())), flush=True)
        # d(term)/d(buffer pixel), after antialias
        if 'stacked' in cap:
            gs = cap['stacked'].cpu().double()
            for k, gb in zip(base, go_buf):
                if gb is None or k not in layout:
                    continue
                c0, nc = layout[k]
                a, b = gs[..., c0:c0 + nc], gb.double()
                nc_ = min(a.shape[-1], b.shape[-1])
                a, b = a[..., :nc_], b[..., :nc_]
                e = (a - b).abs().sum(-1)
                if float(b.abs().max()) == 0 and float(a.abs().max()) == 0:
                    continue
                top = torch.topk(e.reshape(-1), 6).indices
                H = e.shape[1]
                rows = []
                for i in top.tolist():
                    y, x = divmod(i % (H * e.shape[2]), e.shape[2])
                    rows.append({'yx': (y, x), 'err': '%.3e' % float(e[0, y, x]), 'g_gpu': ['%.3e' % v for v in a[0, y, x].tolist()], 'g_or': ['%.3e' % v for v in b[0, y, x].tolist()],
                                 'len_gn': '%.3e' % float(len_o[0, y, x]), 'alpha': '%.4f' % float(a_o[0, y, x]), 'id': int(rast_p[0, y, x, 3])})
                print('  d%s/d%s per pixel: L1 diff %.3e of L1 %.3e; max|g| %.3e; pixels with err > 1e-3 max: %d; in thin-normal pixels: L1 diff %.3e' % (
                    term, k, float(e.sum()), float(b.abs().sum()), float(b.abs().max()), int((e > 1e-3 * b.abs().max()).sum()), float(e[thin].sum())))
                for row in rows:
                    print('      ', json.dumps(row))
            sys.stdout.flush()
    out = {}
    for name, a, b, c in zip(names, tot_p, tot_o, tot_o64):
        if a is not None and b is not None:
            out[name] = ['%.1e' % v for v in rel(a, b)] + ['|'] + ['%.1e' % v for v in rel(a, c)] + ['|'] + ['%.1e' % v for v in rel(b, c)]
    print('SUM-OF-TERMS [gpu vs oracle32 | gpu vs float64 | oracle32 vs float64] (max-norm, L2)', json.dumps(out), flush=True)
    del sc, st, ro, st64, ro64
    if DEV == 'cuda':
        torch.cuda.empty_cache()
