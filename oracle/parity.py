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


def _rel(a, b):
    """(max-norm, L2) error of a against b, relative to b's max / norm"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)), float((a - b).norm() / max(float(b.norm()), 1e-30))


def scene_tick_parity(sc, iteration=10, seed=0):
    """One tick_init of `sc` on its device and the oracle tick on the same state; returns (report, timing).

    Backward on both sides: the config's own total (train.py:718; `msk_loss` for the mask-only config) -- timed on the oracle side as the
    CPU baseline of that config -- and then, for the mask-only config, `reg_loss` on top (untimed), so that the eikonal and sdf_reg gradients
    of the SDF network are covered at that size too.

    report: mesh_faces_equal (bit-exact triangle indices), mesh_verts / mesh_faces, raster_ids_differ (pixels), loss terms of both sides
    with max_rel_loss_diff, and per gradient group -- the 16 tensors of the SDF network, deform, msdf, trans, the grid table, the three
    texture-MLP weights -- the worst max-norm and L2 error relative to the oracle's gradient (max_rel_grad_diff, l2_rel_grad_diff)."""
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

    st = state_from_scene(sc, bg, pts, iteration)
    t0 = time.time()
    ro = OTK.tick_init(st, buffers=base, draws=draws, keep=True)
    t1 = time.time()
    ro['total'].backward(retain_graph=mask_only)
    t2 = time.time()
    if mask_only:
        ro['reg_loss'].backward()
    ref = oracle_grads(st)

    faces_o = ro['_mesh']['faces']
    rast_o = ro['_stages']['rast']
    rep = {'config': f'{nF} frame(s), {g.verts.shape[0]} grid vertices / {g.indices.shape[0]} tets, {H}x{H}, loss set "{sc.loss_set}", '
                     f'{0 if pts is None else pts.shape[0]} eikonal samples',
           'mesh_verts': int(ro['_mesh']['verts'].shape[0]), 'mesh_faces': int(faces_o.shape[0]),
           'mesh_faces_equal': bool(faces_p.shape == faces_o.shape and torch.equal(faces_p, faces_o)),
           'raster_ids_differ': int((rast_p[..., 3] != rast_o[..., 3]).sum()), 'pixels': int(nF * H * H)}
    keys = ('msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss') if mask_only else \
        ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss') + (('ssim_loss',) if st['ssim_weight'] else ())
    losses, worst = {}, 0.0
    for k in keys:
        a, b = float(r[k].detach()), float(ro[k].detach())
        losses[k] = {'gpu': a, 'oracle': b}
        worst = max(worst, abs(a - b) / max(1e-3, abs(b)))
    rep['losses'], rep['max_rel_loss_diff'] = losses, worst
    groups = {'sdf_net': [k for k in ref if k.startswith('sd.')], 'deform': ['deform'], 'msdf': ['msdf'], 'trans': ['trans'], 'table': ['table'],
              'tex_mlp': ['w1', 'w2', 'w3']}
    mx, l2 = {}, {}
    for name, ks in groups.items():
        em, el = 0.0, 0.0
        for k in ks:
            if ref[k] is None:
                continue
            if got[k] is None:
                em = el = float('inf')
                continue
            a, b = _rel(got[k], ref[k])
            em, el = max(em, a), max(el, b)
        mx[name], l2[name] = em, el
    rep['max_rel_grad_diff'], rep['l2_rel_grad_diff'] = mx, l2
    return rep, {'forward_s': t1 - t0, 'backward_s': t2 - t1}
