"""Whole-tick parity harness (TEST INFRASTRUCTURE, like the rest of oracle/): one training tick of a d3h.scene.Scene on the GPU against
the oracle chain (oracle/tick.py, pinned by the reference goldens) on the SAME state -- same parameters, targets, background, surface
samples and shading jitter -- at any size, including BASELINE's.

Used by tests/ (tests/e2e_cases.py, tests/test_gpu_fullsize.py) and by bench.py's `cpu_baseline` leg, which times the oracle tick anyway
and now keeps its losses and gradients instead of throwing them away: `cpu_baseline.parity` in the bench line.  Nothing here is imported
by the product package.

The oracle renders with its OWN rasteriser here (no decision is shared): a pixel whose centre sits within fp32 rounding of a shared edge,
or whose two nearest surfaces z-fight within an ulp, may be won by a different triangle (counted: `raster_ids_differ`), and every such
pixel moves the loss by O(1 / pixels) -- the loss bars below are sized for a handful of them.
"""
import contextlib
import time

import numpy as np
import torch

BBOX = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)          # render/mlptexture.py:94


@contextlib.contextmanager
def fixed_surface_samples(pts):
    """kaolin.ops.mesh.sample_points -> the given pre-drawn points (the eikonal term detaches them, hmsdf.py:858)"""
    import kaolin
    old = kaolin.ops.mesh.sample_points
    if isinstance(pts, (list, tuple)):             # one point set per call, in call order (tick_split x {cloth, body})
        queue = list(pts)
        kaolin.ops.mesh.sample_points = lambda v, f, n, *a, **k: (queue.pop(0)[None], None)
    else:
        kaolin.ops.mesh.sample_points = lambda v, f, n, *a, **k: (pts[None], None)
    try:
        yield
    finally:
        kaolin.ops.mesh.sample_points = old


@contextlib.contextmanager
def recorded_surface_samples(store):
    """kaolin.ops.mesh.sample_points runs as it is; the points of every call are appended to `store`"""
    import kaolin
    old = kaolin.ops.mesh.sample_points

    def rec(*a, **k):
        out = old(*a, **k)
        store.append(out[0][0].detach().clone())
        return out
    kaolin.ops.mesh.sample_points = rec
    try:
        yield
    finally:
        kaolin.ops.mesh.sample_points = old


@contextlib.contextmanager
def fixed_render_draws(draws_list, dev):
    """render.render.render_mesh -> the same function with `_rng_draws` taken from `draws_list` (one dict of 'noise' / 'offset' /
    'pos_noise' per call, as oracle.render.draw_jitter draws them from the seeded CPU generator in the reference's order)"""
    from render import render as R
    old = R.render_mesh
    queue = [{k: v.to(dev) for k, v in d.items()} for d in draws_list]

    def patched(*a, **k):
        k['_rng_draws'] = queue.pop(0)
        return old(*a, **k)
    R.render_mesh = patched
    try:
        yield
    finally:
        R.render_mesh = old


def state_from_scene(sc, background, sampled_pts, iteration):
    """snapshot of a d3h.scene.Scene (the synthetic benchmark scene) as an oracle state: same parameters, same batch"""
    g, F = sc.geometry, sc.FLAGS
    C = lambda t: t.detach().cpu().clone()
    leaf = lambda t: C(t).requires_grad_(True)
    md = F.smplx_model_dict
    body = {k: torch.from_numpy(np.asarray(md[k])) for k in ('v_template', 'J_regressor', 'shapedirs', 'expr_dirs', 'parents', 'weights')}
    tex = sc.material['kd_ks']
    omin, omax = tex._range_host()
    nF = sc.n_frames
    return {'verts': C(g.verts), 'indices': C(g.indices), 'deform': leaf(g.deform), 'msdf': leaf(g.msdf), 'max_disp': g.max_displacement,
            'sd': {k: leaf(v) for k, v in g.sdf_net.state_dict().items()}, 'body': body, 'tmpl': C(g.smplx_deform.vs_template[0]),
            'A0': C(g.smplx_deform.init_A[0]), 'shape': C(F.shape_param), 'expr': C(F.expr_optim), 'root_pose': C(F.root_pose_optim),
            'body_pose': C(F.body_pose_optim), 'jaw_pose': C(F.jaw_pose_optim), 'trans': leaf(F.trans_optim), 'mvp': C(sc.mvp), 'campos': C(sc.campos),
            'res': (sc.res, sc.res),
            'material': {'table': leaf(tex.encoder.params), 'w1': leaf(tex.net.net[0].weight), 'w2': leaf(tex.net.net[2].weight),
                         'w3': leaf(tex.net.net[4].weight), 'bbox': BBOX, 'omin': list(omin), 'omax': list(omax)},
            'all_img': C(sc.all_img), 'all_normal': C(sc.all_normal), 'background': C(background),
            'sampled_pts': C(sampled_pts) if sampled_pts is not None else None, 'iteration': iteration, 'n_iter': F.iter,
            'sdf_regularizer': F.sdf_regularizer, 'eikonal_scale': F.eikonal_scale, 'ssim_weight': F.ssim_weight,
            'loss_set': 'mask' if sc.loss_set == 'mask' else 'full', 'frames': list(range(nF))}


def scene_grads(sc):
    g, tex, F = sc.geometry, sc.material['kd_ks'], sc.FLAGS
    out = {('sd.' + k): p.grad for k, p in g.sdf_net.named_parameters()}
    out.update({'deform': g.deform.grad, 'msdf': g.msdf.grad, 'trans': F.trans_optim.grad, 'table': tex.encoder.params.grad})
    for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
        out[k] = tex.net.net[i].weight.grad
    return out


def oracle_grads(st):
    out = {('sd.' + k): p.grad for k, p in st['sd'].items()}
    out.update({'deform': st['deform'].grad, 'msdf': st['msdf'].grad, 'trans': st['trans'].grad})
    m = st['material']
    out.update({'table': m['table'].grad, 'w1': m['w1'].grad, 'w2': m['w2'].grad, 'w3': m['w3'].grad})
    return out


def _relu_kinks(st, oracle_out, eps=4e-6):
    """covered pixels at which a hidden unit of the texture MLP (mlptexture.py:18-41: two ReLU layers) has a pre-activation within rounding
    of zero: its gate, hence d(colour)/d(position) of that pixel, depends on the summation order.  (With the encoding table at its
    initial +-1e-4 amplitude that is a sizeable share of all pixels; a fitted texture has a few per 10^5.)"""
    from oracle import texmlp as OT
    S, m = oracle_out['_stages'], st['material']
    with torch.no_grad():
        x = S['gb_pos_orig'][S['rast'][..., 3] > 0]
        b0, b1 = torch.tensor(m['bbox'][:3]), torch.tensor(m['bbox'][3:])
        enc = OT.grid_encode(torch.clamp((x - b0) / (b1 - b0), 0, 1), m['table'].detach())
        h1 = enc @ m['w1'].detach().t()
        h2 = torch.relu(h1) @ m['w2'].detach().t()
        return int(((h1.abs() < eps).any(-1) | (h2.abs() < eps).any(-1)).sum())


def _kink_pixels_relu(st, oracle_out, eps=4e-6):
    """(b, y, x) of the covered pixels _relu_kinks counts"""
    from oracle import texmlp as OT
    S, m = oracle_out['_stages'], st['material']
    with torch.no_grad():
        cov = S['rast'][..., 3] > 0
        x = S['gb_pos_orig'][cov]
        b0, b1 = torch.tensor(m['bbox'][:3]), torch.tensor(m['bbox'][3:])
        enc = OT.grid_encode(torch.clamp((x - b0) / (b1 - b0), 0, 1), m['table'].detach())
        h1 = enc @ m['w1'].detach().t()
        h2 = torch.relu(h1) @ m['w2'].detach().t()
        bad = (h1.abs() < eps).any(-1) | (h2.abs() < eps).any(-1)
        return torch.nonzero(cov)[bad]


def kink_grid_vertices(v_def, mesh_verts, faces, rast_ids, pix, neighbours):
    """Boolean mask over the grid vertices whose gradient a discrete decision at the pixels `pix` ([k, 3] rows of (b, y, x)) can move:
    the triangles covering those pixels (and, for antialias decisions, their 4-neighbours: antialias works on pixel PAIRS), their mesh
    vertices, and every grid vertex within one cell diagonal of such a mesh vertex in the deformed grid (a mesh vertex lies on a grid
    edge, hmsdf.py:433 / gshell_tets.py:296-300).  Returns (mask, number of triangles)."""
    N = v_def.shape[0]
    mask = torch.zeros(N, dtype=torch.bool)
    if pix is None or len(pix) == 0:
        return mask, 0
    B, H, W = rast_ids.shape
    tri = []
    for b, y, x in pix.tolist():
        nb = [(y, x)] + ([(y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)] if neighbours else [])
        for yy, xx in nb:
            if 0 <= yy < H and 0 <= xx < W:
                t = int(rast_ids[b, yy, xx]) - 1
                if t >= 0:
                    tri.append(t)
    tri = sorted(set(tri))
    if not tri:
        return mask, 0
    mv = torch.unique(faces[torch.tensor(tri)].reshape(-1))
    p = mesh_verts.detach()[mv].double()
    side = round(N ** (1.0 / 3.0))
    h = float((v_def.max(0).values - v_def.min(0).values).max()) / max(side - 1, 1)
    vd = v_def.detach().double()
    for i in range(0, p.shape[0], 256):
        d = torch.cdist(p[i:i + 256], vd)
        mask |= (d <= 1.8 * h).any(0)
    return mask, len(tri)


def _rel(a, b):
    """(max-norm, L2) error of a against b, relative to b's max / norm"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)), float((a - b).norm() / max(float(b.norm()), 1e-30))


def _to_dtype(o, dt):
    """a state / draws structure with every floating tensor cast to `dt` (leaves stay leaves)"""
    if torch.is_tensor(o):
        if o.is_floating_point():
            t = o.detach().to(dt)
            return t.requires_grad_(True) if o.requires_grad else t
        return o
    if isinstance(o, dict):
        return {k: _to_dtype(v, dt) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_dtype(v, dt) for v in o)
    return o


GRAD_GROUPS = (('sdf_net', lambda k: k.startswith('sd.') and k.endswith('weight')), ('sdf_net_bias', lambda k: k.startswith('sd.') and k.endswith('bias')),
               ('deform', lambda k: k == 'deform'), ('msdf', lambda k: k == 'msdf'), ('trans', lambda k: k == 'trans'), ('table', lambda k: k == 'table'),
               ('tex_mlp', lambda k: k in ('w1', 'w2', 'w3')))


def grad_errors(got, ref, keep=None):
    """per gradient group: (worst max-norm error, worst relative L2 error) of `got` against `ref` (dicts of tensors; `ref` may be float64),
    relative to the reference tensor's max / norm; one-element tensors of a group (the head bias) against the largest entry of the group.
    keep: boolean mask over the rows of the per-grid-vertex tensors (deform, msdf) -- the error is taken over the kept rows, the scale over all"""
    mx, l2 = {}, {}
    for name, sel in GRAD_GROUPS:
        ks = [k for k in ref if sel(k) and ref[k] is not None]
        if not ks:
            mx[name] = l2[name] = None
            continue
        em = el = 0.0
        scale1 = max([float(ref[j].abs().max()) for j in ks] + [1e-30])
        for k in ks:
            if got.get(k) is None:
                em = el = float('inf')
                continue
            a, b = got[k].detach().cpu().double(), ref[k].detach().cpu().double()
            if not bool(torch.isfinite(a).all()) or not bool(torch.isfinite(b).all()):
                em = el = float('inf')
                continue
            bm, bn = max(float(b.abs().max()), 1e-30), max(float(b.norm()), 1e-30)
            if keep is not None and name in ('deform', 'msdf'):
                a, b = a[keep], b[keep]
            if b.numel() == 1 and len(ks) > 1:
                e1 = e2 = float((a - b).abs().max()) / scale1
            else:
                e1, e2 = float((a - b).abs().max()) / bm, float((a - b).norm()) / bn
            em, el = max(em, e1), max(el, e2)
        mx[name], l2[name] = em, el
    return mx, l2


def scene_tick_parity(sc, iteration=10, seed=0, detail=False, share_raster=True, truth64=False):
    """One tick_init of `sc` on its device and the oracle tick on the same state; returns (report, timing).

    Backward on both sides: the config's own total (train.py:718; `msk_loss` for the mask-only config) -- timed on the oracle side as the
    CPU baseline of that config -- and then, for the mask-only config, `reg_loss` on top (untimed), so that the eikonal and sdf_reg gradients
    of the SDF network are covered at that size too.

    report: mesh_faces_equal (bit-exact triangle indices), mesh_verts / mesh_faces, the discrete differences of the two rasterisers
    (raster_ids_differ, alpha_pixels_differ), and two comparisons -- `own_raster` (nothing shared) and `shared_raster` (the oracle renders
    the product's per-pixel winners) -- each with the loss terms of both sides, max_rel_loss_diff, and per gradient group (the 16 tensors of
    the SDF network, deform, msdf, trans, the grid table, the three texture-MLP weights) the worst max-norm and L2 error relative to the
    oracle's gradient (max_rel_grad_diff, l2_rel_grad_diff; None where the loss set gives the oracle no gradient).

    truth64: a third oracle run, the shared-raster one IN FLOAT64 (same formulas, same discrete decisions: the product's winners, the float32
    signs), as a reference for the reference.  report['float64'] then holds, per gradient group, the error of the GPU tick and the error of
    the float32 oracle tick against it -- which of the two a GPU-vs-oracle difference belongs to."""
    from oracle import tick as OTK, render as ORD
    g = sc.geometry
    dev = sc.device
    nF, H = sc.n_frames, sc.res
    gen = torch.Generator().manual_seed(1000 + seed)
    bg = torch.rand(nF, H, H, 3, generator=gen)
    torch.manual_seed(2000 + seed)
    draws = ORD.draw_jitter(nF, H, H)
    mask_only = sc.loss_set == 'mask'
    base = ('shaded',) if mask_only else ('shaded', 'geometric_normal', 'msdf_image')
    save_buf = sc.FLAGS.render_buffers
    sc.FLAGS.render_buffers = base + ('_rast',)
    pts_store = []
    try:
        sc._zero_grad()
        with recorded_surface_samples(pts_store), fixed_render_draws([draws], dev):
            r = g.tick_init(sc.glctx, sc.target(bg.to(dev)), None, sc.material, sc.loss_fn, iteration, None)
    finally:
        sc.FLAGS.render_buffers = save_buf
    if mask_only:
        total = r['msk_loss'] + r['reg_loss']
    elif 'd3h_total' in r:
        total = r['d3h_total']
    else:
        total = r['reg_loss'] + r['normal_loss'] + r['msk_loss'] + r.get('ssim_loss', 0.0)
    total.backward()
    d = g.last_mesh_dict
    faces_p = d['imesh'].t_pos_idx.detach().cpu().long()
    rast_p = d['buffers']['_rast'].detach().cpu()
    got = {k: (None if v is None else v.detach().cpu().clone()) for k, v in scene_grads(sc).items()}
    pts = pts_store[0] if pts_store else None

    def oracle_run(share, dtype=torch.float32):
        st = state_from_scene(sc, bg, pts, iteration)
        kw = {'rast_zw': rast_p[..., 2], 'rast_ids': rast_p[..., 3]} if share else {}
        dr = draws
        if dtype != torch.float32:
            st, dr, kw = _to_dtype(st, dtype), _to_dtype(draws, dtype), _to_dtype(kw, dtype)
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            t0 = time.time()
            ro = OTK.tick_init(st, buffers=base, draws=dr, keep=True, **kw)
            t1 = time.time()
            ro['total'].backward(retain_graph=mask_only)
            t2 = time.time()
            if mask_only:
                ro['reg_loss'].backward()
        finally:
            torch.set_default_dtype(old)
        return st, ro, oracle_grads(st), (t1 - t0, t2 - t1)

    keys = ('msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss') if mask_only else \
        ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss') + (('ssim_loss',) if sc.FLAGS.ssim_weight else ())

    def compare(ro, ref, excl=None):
        """`excl`: boolean mask over the grid vertices excluded from the per-vertex tensors (deform, msdf) in the `*_excl` figures --
        the vertices behind the triangles on which a counted discrete decision (kink) sits; the other tensors are sums over all pixels
        and are never masked"""
        out = {}
        losses, worst = {}, 0.0
        for k in keys:
            a, b = float(r[k].detach()), float(ro[k].detach())
            losses[k] = {'gpu': a, 'oracle': b}
            worst = max(worst, abs(a - b) / max(1e-3, abs(b)))
        out['losses'], out['max_rel_loss_diff'] = losses, worst
        # the SDF network's weights and biases are reported apart: a bias gradient is the plain SUM of dZ over ~6 10^4 points (grid + eikonal samples),
        # positive and negative terms cancelling to a fraction of their magnitude, whose fp32 summation order differs between the two implementations
        groups = {'sdf_net': [k for k in ref if k.startswith('sd.') and k.endswith('weight')],
                  'sdf_net_bias': [k for k in ref if k.startswith('sd.') and k.endswith('bias')], 'deform': ['deform'], 'msdf': ['msdf'],
                  'trans': ['trans'], 'table': ['table'], 'tex_mlp': ['w1', 'w2', 'w3']}
        mx, l2 = {}, {}
        for name, ks in groups.items():
            if all(ref[k] is None for k in ks):
                mx[name] = l2[name] = None                     # the oracle has no gradient for it in this loss set (e.g. the texture in a mask-only tick)
                continue
            em, el = 0.0, 0.0
            for k in ks:
                if ref[k] is None:
                    continue
                if got[k] is None:
                    em = el = float('inf')
                    continue
                if not bool(torch.isfinite(got[k]).all()) or not bool(torch.isfinite(ref[k]).all()):
                    out.setdefault('non_finite', {})[k] = {'gpu': int((~torch.isfinite(got[k])).sum()), 'oracle': int((~torch.isfinite(ref[k])).sum())}
                    em = el = float('inf')                      # never silently: max(x, nan) would keep x
                    continue
                a, b = _rel(got[k], ref[k])
                if ref[k].numel() == 1 and len(ks) > 1:
                    # a one-element tensor (the head bias: d/d(b7) = the SUM of d(loss)/d(sdf) over all grid vertices, positive and negative
                    # terms cancelling to a fraction of their magnitude) is measured against the largest bias gradient of its group, not against
                    # its own -- ill-conditioned -- value
                    scale = max([float(ref[j].abs().max()) for j in ks if ref[j] is not None and j.endswith('bias')] + [1e-30])
                    a = b = float((got[k].detach().cpu().double() - ref[k].detach().cpu().double()).abs().max()) / scale
                em, el = max(em, a), max(el, b)
                out.setdefault('per_tensor', {})[k] = (a, b, float(ref[k].abs().max()), int(ref[k].numel()))
            mx[name], l2[name] = em, el
        out['max_rel_grad_diff'], out['l2_rel_grad_diff'] = mx, l2
        if excl is not None:
            keep_ = ~excl
            mxe, l2e, outl = dict(mx), dict(l2), {'deform': 0, 'msdf': 0}
            for k in ('deform', 'msdf'):
                if ref[k] is None or got[k] is None or mx[k] is None or mx[k] == float('inf'):
                    continue
                a, b = got[k].detach().cpu().double()[keep_], ref[k].detach().cpu().double()[keep_]
                if float(b.abs().max()) == 0.0 and float(a.abs().max()) == 0.0:
                    mxe[k] = l2e[k] = 0.0
                    continue
                # relative to the WHOLE reference tensor (the kept part of a sparse gradient -- msdf: non-zero only along the open boundary -- may
                # be all zero in the reference while the other side has one tiny entry there: a ratio against the kept part alone is 1e19)
                bf = ref[k].detach().cpu().double()
                mxe[k] = float((a - b).abs().max() / max(float(bf.abs().max()), 1e-30))
                l2e[k] = float((a - b).norm() / max(float(bf.norm()), 1e-30))
                # how many of the kept grid vertices carry an error above 2e-3 of the largest gradient entry: the decisions this harness does not
                # count (a two-sided normal flipped on an edge-on triangle, a clamp on its threshold ...) show up as a handful of isolated vertices
                # with an O(1) relative error, rounding as thousands of vertices with a tiny one -- the test bounds both
                e = (a - b).abs().reshape(a.shape[0], -1).max(1).values
                outl[k] = int((e > 2e-3 * max(float(bf.abs().max()), 1e-30)).sum())
            out['max_rel_grad_diff_excl'], out['l2_rel_grad_diff_excl'], out['vertex_outliers_excl'] = mxe, l2e, outl
            out['excluded_grid_vertices'] = int(excl.sum())
        return out

    # (1) the oracle with its OWN rasteriser: nothing shared.  This run is also the one bench.py times as the CPU baseline.
    st, ro, ref, (fwd_s, bwd_s) = oracle_run(False)
    faces_o = ro['_mesh']['faces']
    rast_o = ro['_stages']['rast']
    a_p = d['buffers']['shaded'][..., 3].detach().cpu()
    a_o = ro['_buffers']['shaded'][..., 3].detach()
    alpha_bad = torch.nonzero((a_p - a_o).abs() > 1e-3)
    rep = {'config': f'{nF} frame(s), {g.verts.shape[0]} grid vertices / {g.indices.shape[0]} tets, {H}x{H}, loss set "{sc.loss_set}", '
                     f'{0 if pts is None else pts.shape[0]} eikonal samples',
           'mesh_verts': int(ro['_mesh']['verts'].shape[0]), 'mesh_faces': int(faces_o.shape[0]),
           'mesh_faces_equal': bool(faces_p.shape == faces_o.shape and torch.equal(faces_p, faces_o)),
           'pixels': int(nF * H * H),
           # discrete decisions on which two correct rasterisers may differ: a pixel centre within rounding of an interior edge (either
           # neighbour wins: harmless), and two surfaces closer in depth than the z/w resolution of the reference's 0.001 / 1000 clip planes
           # (a fold of the fitted surface: whichever wins decides whether antialias sees a silhouette there -- `alpha_pixels_differ`)
           'raster_ids_differ': int((rast_p[..., 3] != rast_o[..., 3]).sum()), 'alpha_pixels_differ': int(alpha_bad.shape[0])}
    # discrete decisions counted above -> the grid vertices behind the triangles they sit on (excluded in the `*_excl` figures only)
    v_def_o, mverts_o = ro['_mesh']['v_def'].detach(), ro['_mesh']['verts'].detach()
    relu_pix = None if mask_only else _kink_pixels_relu(st, ro)          # a mask-only tick never reads the texture: no gate can matter
    rep['relu_kinks'] = 0 if relu_pix is None else int(relu_pix.shape[0])
    ids_o = rast_o[..., 3].long()
    pix_own = torch.cat([torch.nonzero(rast_p[..., 3] != rast_o[..., 3]), alpha_bad], 0)
    ex_a, nt_a = kink_grid_vertices(v_def_o, mverts_o, faces_o, ids_o, pix_own, True)
    ex_b, nt_b = kink_grid_vertices(v_def_o, mverts_o, faces_o, rast_p[..., 3].long(), pix_own, True) if rep['mesh_faces_equal'] else (ex_a, 0)
    ex_r, nt_r = kink_grid_vertices(v_def_o, mverts_o, faces_o, ids_o, relu_pix, False)
    rep['own_raster'] = compare(ro, ref, ex_a | ex_b | ex_r)
    rep['own_raster']['kink_triangles'] = nt_a + nt_b + nt_r
    if detail:
        worst = {}
        for k in ('msdf', 'deform', 'trans'):
            if ref[k] is None or got[k] is None:
                continue
            a, b = got[k].double().reshape(-1), ref[k].detach().double().reshape(-1)
            e = (a - b).abs()
            top = torch.topk(e, min(6, e.numel())).indices
            worst[k] = {'oracle_max': float(b.abs().max()), 'oracle_nonzero': int((b != 0).sum()), 'gpu_nonzero': int((a != 0).sum()),
                        'top': [(int(i), float(a[i]), float(b[i])) for i in top.tolist()]}
        rep['grad_detail'] = worst
    if detail:          # stage-by-stage numbers for hunting a mismatch (tools/dbg/gpu_dbg_parity_steps.py)
        posed_p = d['deform_imesh'].v_pos.detach().cpu()
        posed_o = ro['_mesh']['posed'].detach()
        det = {'posed_max_abs_diff': float((posed_p - posed_o).abs().max()),
               'sdf_max_abs_diff': float((d['sdf'].detach().cpu().reshape(-1) - ro['_mesh']['sdf'].detach().reshape(-1)).abs().max()),
               'alpha_sum_gpu': float(a_p.sum()), 'alpha_sum_oracle': float(a_o.sum()), 'pixels': []}
        for b, y, x in alpha_bad[:12].tolist():
            det['pixels'].append({'byx': (b, y, x), 'alpha_gpu': float(a_p[b, y, x]), 'alpha_oracle': float(a_o[b, y, x]),
                                  'rast_gpu': rast_p[b, y, x].tolist(), 'rast_oracle': rast_o[b, y, x].tolist()})
        rep['detail'] = det
    del st, ro, ref
    v_def_o, mverts_o = v_def_o.clone(), mverts_o.clone()
    # (2) the same with the product's per-pixel winners and z/w handed to the oracle renderer (its rasteriser still computes the
    # barycentrics of those winners itself): everything downstream of the discrete pass, compared strictly
    if share_raster:
        st2, ro2, ref2, _ = oracle_run(True)
        a_o2 = ro2['_buffers']['shaded'][..., 3].detach()
        alpha_bad2 = torch.nonzero((a_p - a_o2).abs() > 1e-3)
        ids_p = rast_p[..., 3].long()
        ex_a2, nt_a2 = kink_grid_vertices(v_def_o, mverts_o, faces_o, ids_p, alpha_bad2, True)
        relu_pix2 = None if mask_only else _kink_pixels_relu(st2, ro2)
        ex_r2, nt_r2 = kink_grid_vertices(v_def_o, mverts_o, faces_o, ids_p, relu_pix2, False)
        rep['shared_raster'] = compare(ro2, ref2, ex_a2 | ex_r2)
        rep['shared_raster']['alpha_pixels_differ'] = int(alpha_bad2.shape[0])
        rep['shared_raster']['kink_triangles'] = nt_a2 + nt_r2
        if detail:          # the worst per-vertex entries of the strict comparison: (flat index, product, oracle) + how many entries carry the error
            worst = {}
            for k in ('msdf', 'deform'):
                if ref2[k] is None or got[k] is None:
                    continue
                a, b = got[k].double().reshape(-1), ref2[k].detach().double().reshape(-1)
                e = (a - b).abs()
                top = torch.topk(e, min(8, e.numel())).indices
                bmax = float(b.abs().max())
                worst[k] = {'oracle_max': bmax, 'entries_above_1e-3_of_max': int((e > 1e-3 * bmax).sum()), 'entries_above_3e-4_of_max': int((e > 3e-4 * bmax).sum()),
                            'top': [(int(i), float(a[i]), float(b[i])) for i in top.tolist()]}
            rep['shared_raster']['grad_detail'] = worst
        if truth64:
            excl2 = ex_a2 | ex_r2
            ref32 = {k: (None if v is None else v.detach().clone()) for k, v in ref2.items()}
            loss32 = {k: float(ro2[k].detach()) for k in keys}
            faces32 = ro2['_mesh']['faces']
            del st2, ro2, ref2
            st3, ro3, ref3, tm3 = oracle_run(True, torch.float64)
            keep3 = ~excl2
            g_mx, g_l2 = grad_errors(got, ref3, keep3)
            o_mx, o_l2 = grad_errors(ref32, ref3, keep3)
            lg = max(abs(float(r[k].detach()) - float(ro3[k].detach())) / max(1e-3, abs(float(ro3[k].detach()))) for k in keys)
            lo = max(abs(loss32[k] - float(ro3[k].detach())) / max(1e-3, abs(float(ro3[k].detach()))) for k in keys)
            rep['float64'] = {'same_mesh': bool(torch.equal(faces32, ro3['_mesh']['faces'])), 'gpu_max': g_mx, 'gpu_l2': g_l2, 'oracle32_max': o_mx,
                              'oracle32_l2': o_l2, 'gpu_loss': lg, 'oracle32_loss': lo, 'forward_s': tm3[0], 'backward_s': tm3[1]}
    return rep, {'forward_s': fwd_s, 'backward_s': bwd_s}
