"""End-to-end parity of one training tick of each stage: the product (geometry.hmsdf.HmSDFTetsGeometry.tick_init / tick_split /
tick_seq on the HIP kernels through the C ABI; `dev` = 'cuda', or 'cpu' only under the test-only emulator hook) against
  (a) tests/golden/tick_{init,split,seq}.npz -- the REFERENCE's own tick_* run in the dev container (tools/gen_golden.py), and
  (b) the oracle chain (oracle/tick.py, pinned by (a)) on seeded states of other sizes / several frames / the default loss stack.
Every loss term and d(total)/d{SDF network, deform, msdf, pose translation, grid table, texture MLP} are compared.
"""
import contextlib
import os

import numpy as np
import torch

from conftest import golden

BBOX = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)


from oracle.parity import fixed_surface_samples, fixed_render_draws, state_from_scene, scene_grads, oracle_grads      # noqa: E402,F401


def build_product(dev, st, grid_res, buffers, normal_loss_fn=None):
    """the product's geometry / material / target / FLAGS carrying exactly the oracle state `st` (oracle.tick's dict).  `buffers`:
    FLAGS.render_buffers* -- 'all' = the reference's 12 buffers, None = tick_*'s default (what the tick reads), or an explicit tuple"""
    from d3h import scene
    from geometry.hmsdf import HmSDFTetsGeometry
    from render.mlptexture import MLPTexture3D
    import nvdiffrast.torch as dr
    D = lambda t: t.detach().clone().to(dev)
    nF = st['trans'].shape[0]
    H, W = st['res']
    F = scene.make_flags(res=H, grid_n=2, n_frames=nF, device=dev, prefit_steps=0, body_verts=512)
    F.train_res = [H, W]
    F.tet_grid = (st['verts'].numpy(), st['indices'].numpy())
    md = {k: v.numpy() for k, v in st['body'].items()}
    md['posedirs'] = np.zeros((54 * 9, md['v_template'].shape[0] * 3), np.float32)          # does not reach the joint transforms
    F.smplx_model_dict = md
    F.shape_param, F.expr_optim = D(st['shape']), D(st['expr'])
    F.body_pose_optim, F.root_pose_optim, F.jaw_pose_optim = D(st['body_pose']), D(st['root_pose']), D(st['jaw_pose'])
    F.trans_optim = D(st['trans']).requires_grad_(True)
    F.sdf_mlp_pretrain_smpl_steps = 0
    F.sdf_init_fn = lambda x: torch.zeros(x.shape[0], device=x.device)
    F.iter, F.sdf_regularizer, F.eikonal_scale = st['n_iter'], st['sdf_regularizer'], st.get('eikonal_scale')
    F.ssim_weight = st.get('ssim_weight', 0.0)
    F.render_buffers = F.render_buffers_split = F.render_buffers_seq = buffers
    F.normal_loss_fn = normal_loss_fn
    F.visualize_watertight = False
    for k, v in (st.get('flags') or {}).items():
        if k != 'grid_res':
            setattr(F, k, v)
    if 'seq' in st:
        F.use_nonrigid_deform, F.deform_checkpoint, F.sdf_deform_pretrain_steps = True, None, 0
    g = HmSDFTetsGeometry(grid_res, 1.0, F)
    assert abs(g.max_displacement - st['max_disp']) < 1e-12
    g.sdf_net.load_state_dict({k: D(v) for k, v in st['sd'].items()})
    g.smplx_deform.vs_template = D(st['tmpl'])[None]
    g.smplx_deform.init_A = D(st['A0'])[None]
    g.smplx_deform._knn_grid = None
    with torch.no_grad():
        g.msdf.copy_(D(st['msdf']))
        g.deform.copy_(D(st['deform']))
    m = st['material']
    tex = MLPTexture3D(g.getAABB(), channels=6, min_max=[torch.tensor(m['omin'], device=dev), torch.tensor(m['omax'], device=dev)]).to(dev)
    with torch.no_grad():
        tex.encoder.params.copy_(D(m['table']))
        for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
            tex.net.net[i].weight.copy_(D(m[k]))
    mat = {'kd_ks': tex, 'bsdf': 'pbr'}
    frames = st.get('frames') or list(range(st['mvp'].shape[0]))
    target = {'idx': frames, 'mvp': D(st['mvp']), 'campos': D(st['campos']), 'resolution': [H, W], 'spp': 1, 'background': D(st['background']),
              'all_img': D(st['all_img']), 'all_normal': D(st['all_normal']) if st.get('all_normal') is not None else None}
    for k in ('cloth_img', 'cloth_normal', 'body_img', 'body_normal'):
        if k in st:
            target[k] = D(st[k])
    if 'seq' in st:
        sq = st['seq']
        g.nonrigid.load_state_dict({k: D(v) for k, v in sq['nr_sd'].items()})
        with torch.no_grad():
            g.fix_code.copy_(D(sq['fix_code']))
        F.v_labels, F.face_labels = D(sq['v_labels']), D(sq['face_labels'])
        F.connected_faces, F.edges, F.body_f = D(sq['connected_faces']), D(sq['edges']), D(sq['body_f'])
        g._init_basedeform(D(sq['base_v']), D(sq['base_f']), D(sq['body_v']), D(sq['cloth_v']))

    def loss_fn(img, ref):
        from render import renderutils as ru
        return ru.image_loss(img, ref, loss='l1', tonemapper='log_srgb')                      # train.py:81 'logl1'
    if os.environ.get('D3H_TEST_PLAIN_LOSS') != '1':          # '1': a bare callable, as train.py's createLoss builds it (recognised by probing)
        loss_fn.d3h_spec = ('l1', 'log_srgb')
    return {'geometry': g, 'material': mat, 'target': target, 'FLAGS': F, 'loss_fn': loss_fn, 'glctx': dr.RasterizeGLContext(), 'tex': tex}


def product_tick(P, st, dev):
    g = P['geometry']
    with fixed_surface_samples(st['sampled_pts'].to(dev) if st.get('sampled_pts') is not None else None):
        r = g.tick_init(P['glctx'], P['target'], None, P['material'], P['loss_fn'], st['iteration'], None)
    if st.get('loss_set', 'full') == 'mask':
        total = r['msk_loss']
    else:
        total = r['reg_loss'] + r['normal_loss'] + r['msk_loss'] + (r['ssim_loss'] if 'ssim_loss' in r else 0.0)
    return r, total


def product_grads(P):
    g, tex, F = P['geometry'], P['tex'], P['FLAGS']
    out = {('sd.' + k): p.grad for k, p in g.sdf_net.named_parameters()}
    out.update({'deform': g.deform.grad, 'msdf': g.msdf.grad, 'trans': F.trans_optim.grad, 'table': tex.encoder.params.grad})
    for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
        out[k] = tex.net.net[i].weight.grad
    return out


def _cmp_grads(got, ref, tol, what, floor=1e-7, kinks=0):
    """Per tensor: max |a - b| <= tol * max|b|; tensors whose reference gradient is identically zero must be (numerically) zero.
    `kinks` = number of places where the tick sits within fp32 rounding of a kink of the (piecewise smooth) function it evaluates, as
    counted by image_flips() and relu_kinks() below: there the derivative is discontinuous and which side an implementation lands on is
    decided by its summation order, so one pixel's contribution (a few per cent of the gradient of one triangle's vertices, and of
    whatever they feed) legitimately differs.  With kinks > 0 the criterion is the relative L2 error <= 10 tol per tensor."""
    worst = {}
    for k, b in ref.items():
        a = got[k]
        if b is None:
            assert a is None or float(a.abs().max()) <= floor, (what, k, 'expected no gradient')
            continue
        assert a is not None, (what, k, 'missing gradient')
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        den = float(b.abs().max())
        err = (a - b).abs()
        worst[k] = float(err.max()) / max(den, 1e-30)
        if kinks == 0:
            assert float(err.max()) <= tol * den + floor, (what, k, float(err.max()), den)
        else:
            l2 = float((a - b).norm() / (b.norm() + 1e-30))
            assert l2 <= 10 * tol, (what, k, 'kinks', kinks, 'l2', l2, float(err.max()), den)
    return worst


def relu_kinks(st, oracle_out, eps=4e-6):
    """covered pixels at which a hidden unit of the texture MLP (mlptexture.py:18-41: two ReLU layers) has a pre-activation within
    rounding of zero: its gate, hence d(colour)/d(position) of that pixel, depends on the summation order"""
    from oracle import texmlp as OT
    S, m = oracle_out['_stages'], st['material']
    with torch.no_grad():
        x = S['gb_pos_orig'][S['rast'][..., 3] > 0]
        b0, b1 = torch.tensor(m['bbox'][:3]), torch.tensor(m['bbox'][3:])
        enc = OT.grid_encode(torch.clamp((x - b0) / (b1 - b0), 0, 1), m['table'].detach())
        h1 = enc @ m['w1'].detach().t()
        h2 = torch.relu(h1) @ m['w2'].detach().t()
        return int(((h1.abs() < eps).any(-1) | (h2.abs() < eps).any(-1)).sum())


def image_flips(P, oracle_out, thresh=1e-3):
    """number of pixels at which the product's and the oracle's antialiased, channel-concatenated images differ by more than thresh"""
    a = P['geometry'].last_mesh_dict['buffers']['_stacked'].detach().cpu()
    b = oracle_out['_stages']['post_aa'].detach()
    assert a.shape == b.shape, (a.shape, b.shape)
    return int(((a - b).abs().max(-1).values > thresh).sum())


def check_tick_init_golden(dev, loss_tol=2e-4, grad_tol=2e-3, buffers=None):
    """product tick_init == the REFERENCE's tick_init (golden), incl. the MobileNetV2-feature normal loss (seeded random trunk)"""
    from geometry.perceptual import MobileNetPerceptualLoss
    from oracle import tick as OTK
    g = dict(golden('tick_init.npz'))
    st = OTK.state_from_golden(g)
    nfn = MobileNetPerceptualLoss(use_gpu=(dev != 'cpu'), seed=int(g['trunk_seed']))
    if dev != 'cpu':
        nfn = nfn.to(dev)
    P = build_product(dev, st, int(g['grid_res']), buffers, normal_loss_fn=nfn)      # None: tick_init's default = the three buffers it reads
    r, total = product_tick(P, st, dev)
    d = P['geometry'].last_mesh_dict
    assert d['deform_imesh'].v_pos.shape[-2] == int(g['n_mesh_verts']) and d['deform_imesh'].t_pos_idx.shape[0] == int(g['n_mesh_faces'])
    for k in ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss'):
        a, b = float(r[k]), float(g['loss.' + k])
        assert abs(a - b) <= loss_tol * max(1e-3, abs(b)), (k, a, b)
    assert abs(float(total) - float(g['loss.total'])) <= loss_tol * float(g['loss.total'])
    total.backward()
    ref = {k[5:]: torch.from_numpy(g[k]) for k in g if k.startswith('grad.')}
    return _cmp_grads(product_grads(P), ref, grad_tol, 'tick_init golden')


def _golden_state(name, sd_name='tick_init.npz'):
    from oracle import tick as OTK
    g = dict(golden(name))
    if not any(k.startswith('sd.') for k in g):             # the split / seq goldens share the SDF network of tick_init.npz
        base = golden(sd_name)
        g.update({k: v for k, v in base.items() if k.startswith('sd.')})
        for k in ('verts', 'indices', 'deform', 'msdf'):     # tick_seq.npz: the tet grid takes no part, any grid serves the constructor
            if k not in g:
                g[k] = base[k]
    if 'nr_sd_from' in g:                                    # tick_seq.npz: the non-rigid network is the one stored in seq.npz
        sq = golden(str(g['nr_sd_from']))
        g.update({'seq.nr_sd.' + k[6:]: sq[k] for k in sq.files if k.startswith('nr_sd.')})
    return g, OTK.state_from_golden(g)


def split_draws(g, n_calls, dev='cpu'):
    """the jitter draws of `n_calls` consecutive render_mesh calls after torch.manual_seed(draws_seed) (CPU generator, reference order)"""
    from oracle import render as ORD
    res = int(g['res'])
    torch.manual_seed(int(g['draws_seed']))
    return [ORD.draw_jitter(1, res, res) for _ in range(n_calls)]


def product_tick_split(P, st, dev, draws, pts, crop_seed, lpips=False):
    """tick_split x {cloth, body} and the total of train.py:1087, as Scene.step_split / the reference loop form it"""
    import random
    g = P['geometry']
    out, total = {}, 0.0
    random.seed(crop_seed)
    with fixed_surface_samples([p.to(dev) for p in pts]), fixed_render_draws(draws, dev):
        for typ in ('cloth', 'body'):
            r = g.tick_split(P['glctx'], P['target'], None, P['material'], P['loss_fn'], st['iteration'], None, type=typ)
            out[typ] = r
            total = total + r['img_loss'] + r['normal_loss'] + r['reg_loss'] + 10 * r['msk_loss']
    return out, total


SPLIT_KEYS = ('img_loss', 'msk_loss', 'depth_loss', 'sdf_reg_loss', 'eik_loss', 'mesh_msdf_reg_loss', 'monochrome_loss', 'mtl_smooth_loss',
              'chroma_loss', 'delta_loss', 'reg_loss', 'geo_reg_loss', 'shading_reg_loss', 'normal_loss_mse', 'normal_loss_cos', 'normal_loss')


def check_tick_split_golden(dev, loss_tol=2e-4, grad_tol=5e-4, buffers='all'):
    """product tick_split x {cloth, body} == the REFERENCE's (tests/golden/tick_split.npz): 16 loss terms per type, the total of
    train.py:1087 and d(total)/d{SDF net, deform, msdf, trans, table, w1-3}; MobileNetV2 normal loss on the seeded 448-crop"""
    from geometry.perceptual import MobileNetPerceptualLoss
    g, st = _golden_state('tick_split.npz')
    nfn = MobileNetPerceptualLoss(use_gpu=(dev != 'cpu'), seed=int(g['trunk_seed']))
    if dev != 'cpu':
        nfn = nfn.to(dev)
    P = build_product(dev, st, int(g['grid_res']), buffers, normal_loss_fn=nfn)
    P['FLAGS'].share_sdf_sweep = True                       # what Scene.step_split runs: one sweep for both extractions
    P['geometry']._sweep_cache = None
    rs, total = product_tick_split(P, st, dev, split_draws(g, 2), [st['sampled_pts.cloth'], st['sampled_pts.body']], int(g['crop_seed']))
    for typ in ('cloth', 'body'):
        for k in SPLIT_KEYS:
            a, b = float(rs[typ][k]), float(g[f'loss.{typ}.{k}'])
            assert abs(a - b) <= loss_tol * max(1e-3, abs(b)), (typ, k, a, b)
    assert abs(float(total) - float(g['loss.total'])) <= loss_tol * float(g['loss.total'])
    total.backward()
    ref = {k[5:]: torch.from_numpy(g[k]) for k in g if k.startswith('grad.')}
    return _cmp_grads(product_grads(P), ref, grad_tol, 'tick_split golden')


SEQ_KEYS = ('all_img_loss', 'all_msk_loss', 'cloth_img_loss', 'cloth_msk_loss', 'body_img_loss', 'body_msk_loss', 'laplacian_loss',
            'mtl_smooth_loss', 'chroma_loss', 'delta_loss', 'reg_loss', 'shading_reg_loss', 'normal_loss', 'colli_loss', 'nds_normal_loss')


def seq_totals(r):
    """train.py:1412-1421 -> (image-driven part, total)"""
    img_part = 250 * r['normal_loss'] + 0.1 * r['reg_loss'] + (r['body_msk_loss'] + r['cloth_msk_loss'] + r['all_msk_loss'])
    return img_part, img_part + 1000000 * r['laplacian_loss'] + 100000 * r['colli_loss'] + 1000 * r['nds_normal_loss'] + r['delta_loss']


def seq_params(P):
    g, F = P['geometry'], P['FLAGS']
    return [('nr.' + k, p) for k, p in g.nonrigid.named_parameters()] + [('fix_code', g.fix_code), ('trans', F.trans_optim)]


def check_tick_seq_golden(dev, loss_tol=2e-4, grad_tol=5e-4, buffers='all'):
    """product tick_seq == the REFERENCE's (tests/golden/tick_seq.npz): 15 loss terms, visible triangles, the total of
    train.py:1412-1421, d(total) and d(image-driven part) w.r.t. the non-rigid network, fix_code, trans (+ table, w1-3 of the total)"""
    from geometry.perceptual import MobileNetPerceptualLoss
    g, st = _golden_state('tick_seq.npz')
    nfn = MobileNetPerceptualLoss(use_gpu=(dev != 'cpu'), seed=int(g['trunk_seed']))
    if dev != 'cpu':
        nfn = nfn.to(dev)
    P = build_product(dev, st, int(g['grid_res']), buffers, normal_loss_fn=nfn)
    with fixed_render_draws(split_draws(g, 1), dev):
        r = P['geometry'].tick_seq(P['glctx'], P['target'], None, P['material'], P['loss_fn'], st['iteration'], None, t='all')
    for k in SEQ_KEYS:
        a, b = float(r[k]), float(g['loss.' + k])
        assert abs(a - b) <= loss_tol * max(1e-6, abs(b)), (k, a, b)
    if r.get('visible_triangles') is not None:
        assert torch.equal(r['visible_triangles'].cpu().long(), torch.from_numpy(g['visible_triangles']).long())
    assert (r['delta'].detach().cpu() - torch.from_numpy(g['delta'])).abs().max() < 2e-6
    img_part, total = seq_totals(r)
    assert abs(float(img_part) - float(g['loss.img_part'])) <= loss_tol * float(g['loss.img_part'])
    assert abs(float(total) - float(g['loss.total'])) <= loss_tol * float(g['loss.total'])
    names = seq_params(P)
    gi = torch.autograd.grad(img_part, [p for _, p in names], retain_graph=True, allow_unused=True)
    worst = _cmp_grads({k: x for (k, _), x in zip(names, gi)}, {k[9:]: torch.from_numpy(g[k]) for k in g if k.startswith('grad_img.')}, grad_tol,
                       'tick_seq golden, image-driven part')
    total.backward()
    got = {k: p.grad for k, p in names}
    got['table'] = P['tex'].encoder.params.grad
    for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
        got[k] = P['tex'].net.net[i].weight.grad
    worst.update({'total.' + k: v for k, v in _cmp_grads(got, {k[5:]: torch.from_numpy(g[k]) for k in g if k.startswith('grad.')}, grad_tol,
                                                         'tick_seq golden').items()})
    return worst


def make_state(n=14, res=80, frames=2, seed=0, n_samples=3000, ssim_weight=1.0, loss_set='full', iteration=40, body_verts=512,
               msdf_shift=0.15, deform_amp=0.3):
    """a seeded oracle state of a chosen size: the pre-fitted SDF network of the tick_init golden on a Kuhn grid of n^3 cubes, `frames`
    frames with their own poses / translations, targets displaced from the render"""
    from d3h import synth
    from oracle import tick as OTK, lbs as OL, texmlp as OT
    g = dict(golden('tick_init.npz'))
    gen = torch.Generator().manual_seed(1000 + seed)
    verts, tets = (torch.from_numpy(a) for a in synth.kuhn_grid(n))
    m = synth.make_body_model(n_verts=body_verts, seed=0, n_shape=10, n_expr=5)
    body = {k: torch.from_numpy(v) for k, v in m.items() if k != 'posedirs'}
    leaf = lambda t: t.clone().requires_grad_(True)
    betas = torch.zeros(1, 10)
    bp0 = torch.zeros(1, 63); bp0[:, 2] = torch.pi / 36; bp0[:, 5] = -torch.pi / 36
    z3, z45 = torch.zeros(1, 3), torch.zeros(1, 45)
    J0 = OL.joints_from_shape(body, betas, torch.zeros(1, 5))
    A0 = OL.pose_transforms(body, OL.full_pose(z3, bp0, z3, z3, z3, z45, z45), J0)[0]
    tmpl = OL.blend_apply(body['v_template'], A0, body['weights'], False)                      # posed template (posedirs = 0)
    H = W = res
    mv, mvp, campos = synth.camera(res, dist=3.0)
    table = (torch.rand(2 * OT.grid_layout()[1], generator=gen) * 2 - 1) * 0.3
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    cx, cy, rx, ry = 0.52 * W, 0.47 * H, 0.17 * W, 0.26 * H
    msk = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) < 1).float()[None, ..., None].expand(frames, -1, -1, -1)
    nx, ny = (xx - cx) / rx, -(yy - cy) / ry
    nz = (1 - (nx ** 2 + ny ** 2)).clamp(min=0.05).sqrt()
    nrm = torch.nn.functional.normalize(torch.stack([nx, ny, nz], -1), dim=-1)[None] * msk
    all_img = torch.cat([torch.tensor([0.55, 0.45, 0.40]).expand(frames, H, W, 3) * msk, msk], -1).contiguous()
    st = {'verts': verts, 'indices': tets, 'deform': leaf((torch.rand(verts.shape, generator=gen) * 2 - 1) * deform_amp),
          'msdf': leaf((torch.rand(verts.shape[0], generator=gen) - msdf_shift).clamp(-1, 1)), 'max_disp': 1.0 / (2 * n) * 1.0 / 2.1,
          'sd': {k[3:]: leaf(torch.from_numpy(g[k])) for k in g if k.startswith('sd.')}, 'body': body, 'tmpl': tmpl, 'A0': A0,
          'shape': betas, 'expr': torch.zeros(frames, 5), 'root_pose': 0.05 * torch.randn(frames, 3, generator=gen),
          'body_pose': synth.poses(frames, seed=77 + seed) * 0.5, 'jaw_pose': torch.zeros(frames, 3),
          'trans': leaf(0.02 * torch.randn(frames, 3, generator=gen)),
          'mvp': torch.from_numpy(mvp)[None].expand(frames, -1, -1).contiguous(), 'campos': torch.from_numpy(campos)[None].expand(frames, -1).contiguous(),
          'res': (H, W),
          'material': {'table': leaf(table), 'w1': leaf(torch.from_numpy(g['w1'])), 'w2': leaf(torch.from_numpy(g['w2'])),
                       'w3': leaf(torch.from_numpy(g['w3'])), 'bbox': BBOX, 'omin': g['omin'].tolist(), 'omax': g['omax'].tolist()},
          'all_img': all_img, 'all_normal': nrm.contiguous(), 'background': torch.rand(frames, H, W, 3, generator=gen),
          'iteration': iteration, 'n_iter': 2001, 'sdf_regularizer': 0.2, 'eikonal_scale': None, 'ssim_weight': ssim_weight, 'loss_set': loss_set}
    with torch.no_grad():
        mm = OTK.get_mesh_init(st, list(range(frames)))
        st['sampled_pts'] = OTK.surface_samples(mm['posed'][0], mm['faces'], n_samples, generator=gen) if n_samples else None
    return st


def make_split_state(**kw):
    """make_state + what tick_split reads: distinct garment / body targets (hmsdf.py:936-944), the flags of train.py:1555-1616 (with
    the raised mSDF-regulariser scales of the golden), one set of surface samples per extraction"""
    from oracle import tick as OTK
    st = make_state(**kw)
    H, W = st['res']
    frames = st['mvp'].shape[0]
    gen = torch.Generator().manual_seed(2000 + kw.get('seed', 0))
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')

    def ell(cx, cy, rx, ry, albedo):
        msk = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) < 1).float()[None, ..., None].expand(frames, -1, -1, -1)
        nx, ny = (xx - cx) / rx, -(yy - cy) / ry
        nz = (1 - (nx ** 2 + ny ** 2)).clamp(min=0.05).sqrt()
        nrm = torch.nn.functional.normalize(torch.stack([nx, ny, nz], -1), dim=-1)[None] * msk
        return torch.cat([torch.tensor(albedo).expand(frames, H, W, 3) * msk, msk], -1).contiguous(), nrm.contiguous()
    st['cloth_img'], st['cloth_normal'] = ell(0.52 * W, 0.41 * H, 0.18 * W, 0.17 * H, [0.30, 0.50, 0.65])
    st['body_img'], st['body_normal'] = ell(0.50 * W, 0.48 * H, 0.16 * W, 0.27 * H, [0.60, 0.45, 0.35])
    st['flags'] = dict(use_mesh_msdf_reg=True, msdf_reg_open_scale=2e-5, msdf_reg_close_scale=6e-5, lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025,
                       lambda_chroma=0.05, texture_res=[H, W], grid_res=2 * kw.get('n', 14))
    ns = kw.get('n_samples', 3000)
    with torch.no_grad():
        for typ in ('cloth', 'body'):
            mm = OTK.get_mesh_split(st, list(range(frames)), typ)
            st['sampled_pts.' + typ] = OTK.surface_samples(mm['posed'][0], mm['faces'], ns, generator=gen)
    return st


def check_tick_split_vs_oracle(dev, loss_tol=2e-4, grad_tol=2e-3, lpips_net=None, **kw):
    """the product's DEFAULT split-stage path (what Scene.step_split runs: dead-buffer elimination, fused pixel losses incl. the material
    smoothness term, MSE + cosine normal term, one shared SDF sweep for both extractions, optional LPIPS) against the oracle chain
    (pinned by tick_split.npz) on a seeded state of another size / several frames"""
    import random
    from oracle import tick as OTK, raster as OR, render as ORD
    st = make_split_state(ssim_weight=0.0, **kw)
    H, W = st['res']
    nF = st['mvp'].shape[0]
    buffers = ('shaded', 'geometric_normal', 'msdf_image', 'kd', 'kd_grad', 'ks_grad', 'normal_grad', '_rast')
    P = build_product(dev, st, 2 * kw.get('n', 14), buffers)
    lp_o = None
    if lpips_net is not None:
        import lpips
        lp_o = lpips.LPIPS(net=lpips_net, pretrained=False, trunk_seed=3, verbose=False)
        lp_p = lpips.LPIPS(net=lpips_net, pretrained=False, trunk_seed=3, verbose=False)
        lp_p.load_state_dict(lp_o.state_dict())
        P['FLAGS'].lpips_fn, P['FLAGS'].lpips_weight = lp_p.to(dev), 0.5
        st['lpips_fn'], st['lpips_weight'] = lp_o, 0.5
    P['FLAGS'].share_sdf_sweep = True
    P['geometry']._sweep_cache = None
    torch.manual_seed(77)
    draws = [ORD.draw_jitter(nF, H, W) for _ in range(2)]
    pts = [st['sampled_pts.cloth'], st['sampled_pts.body']]
    g = P['geometry']
    rs, rasts, total = {}, {}, 0.0
    with fixed_surface_samples([p.to(dev) for p in pts]), fixed_render_draws(draws, dev):
        for typ in ('cloth', 'body'):
            r = g.tick_split(P['glctx'], P['target'], None, P['material'], P['loss_fn'], st['iteration'], None, type=typ)
            rs[typ], rasts[typ] = r, g.last_mesh_dict['buffers']['_rast'].detach().cpu()
            total = total + r['img_loss'] + r['normal_loss'] + r['reg_loss'] + 10 * r['msk_loss']
    tot_o, kinks = 0.0, 0
    for typ, dr in zip(('cloth', 'body'), draws):
        rast_p = rasts[typ]
        with torch.no_grad():
            mo = OTK.get_mesh_split(st, list(range(nF)), typ)
            rast_own, _ = OR.rasterize(ORD.xfm_points(mo['posed'], st['mvp']), mo['faces'], H, W)
        id_diff = rast_p[..., 3] != rast_own[..., 3]
        assert int(id_diff.sum()) <= 3, f'{typ}: {int(id_diff.sum())} pixels differ in triangle id from the oracle rasteriser'
        assert (rast_p[..., 2] - rast_own[..., 2])[~id_diff].abs().max() < 2e-6
        ro = OTK.tick_split(st, typ, draws=dr, pts=st['sampled_pts.' + typ], keep=True, rast_zw=rast_p[..., 2], rast_ids=rast_p[..., 3])
        keys = tuple(k for k in SPLIT_KEYS) + (('lpips_loss',) if lpips_net is not None else ())
        for k in keys:
            a = float(rs[typ][k]) if k != 'lpips_loss' else float(g.last_lpips_loss) if typ == 'body' else None
            if a is None:
                continue
            b = float(ro[k])
            assert abs(a - b) <= loss_tol * max(1e-3, abs(b)), (typ, k, a, b)
        tot_o = tot_o + ro['total']
        kinks += relu_kinks(st, ro)
    assert abs(float(total) - float(tot_o)) <= loss_tol * abs(float(tot_o))
    total.backward()
    tot_o.backward()
    worst = _cmp_grads(product_grads(P), oracle_grads(st), grad_tol, 'tick_split vs oracle chain', kinks=kinks)
    worst['_kinks'] = kinks
    return worst


def check_tick_init_vs_oracle(dev, loss_tol=2e-4, grad_tol=2e-3, **kw):
    """the product's DEFAULT path (fused pixel-loss pass + affine loss head, MSE + cosine normal term, SSIM, fused second-order eikonal,
    several frames each with its own pose) against the oracle chain on the same state"""
    from oracle import tick as OTK
    st = make_state(**kw)
    buffers = ('shaded',) if st['loss_set'] == 'mask' else ('shaded', 'geometric_normal', 'msdf_image')
    P = build_product(dev, st, 2 * kw.get('n', 14), buffers + ('_rast',))
    r, total = product_tick(P, st, dev)
    d = P['geometry'].last_mesh_dict
    # Discrete decisions first, on their own: the product's per-pixel winners and z/w against the oracle rasteriser's.  A pixel centre
    # within rounding of a shared edge can be won by either triangle, and antialias decides which of two neighbouring pixels is nearer
    # by comparing z/w values that differ in the last bit between the two rasterisers (the whole body spans ~10 ulps of z/w with the
    # reference's 0.001 / 1000 clip planes).  One such pixel changes the SSIM gradient of its whole 11 x 11 window by per cents, so the
    # oracle then renders WITH the product's winners and z/w: everything downstream of the discrete pass is compared strictly.
    rast_p = d['buffers']['_rast'].detach().cpu()
    with torch.no_grad():
        mo = OTK.get_mesh_init(st, st.get('frames') or list(range(st['mvp'].shape[0])))
        from oracle import raster as OR, render as ORD
        rast_own, _ = OR.rasterize(ORD.xfm_points(mo['posed'], st['mvp']), mo['faces'], st['res'][0], st['res'][1])
    id_diff = rast_p[..., 3] != rast_own[..., 3]
    assert int(id_diff.sum()) <= 3, f'{int(id_diff.sum())} pixels differ in triangle id from the oracle rasteriser'
    assert (rast_p[..., 2] - rast_own[..., 2])[~id_diff].abs().max() < 2e-6
    assert (rast_p[..., 0:2] - rast_own[..., 0:2])[~id_diff].abs().max() < 2e-4
    ro = OTK.tick_init(st, buffers=buffers, keep=True, rast_zw=rast_p[..., 2], rast_ids=rast_p[..., 3])
    ro['total'].backward()
    assert torch.equal(d['imesh'].t_pos_idx.cpu().long(), ro['_mesh']['faces']), 'extracted faces differ from the oracle'
    keys = ('msk_loss',) if st['loss_set'] == 'mask' else ('img_loss', 'msk_loss', 'eik_loss', 'sdf_reg_loss', 'reg_loss', 'normal_loss') + \
        (('ssim_loss',) if st['ssim_weight'] else ())
    for k in keys:
        a, b = float(r[k]), float(ro[k])
        assert abs(a - b) <= loss_tol * max(1e-3, abs(b)), (k, a, b)
    if 'd3h_total' in r and st['loss_set'] != 'mask':
        assert abs(float(r['d3h_total']) - float(ro['total'])) <= loss_tol * abs(float(ro['total']))
        total = r['d3h_total']                                   # what Scene.step() back-propagates
    total.backward()
    flips = image_flips(P, ro)
    assert flips <= 2, f'{flips} pixels of the antialiased image differ from the oracle'
    kinks = flips + relu_kinks(st, ro)
    worst = _cmp_grads(product_grads(P), oracle_grads(st), grad_tol, 'tick_init vs oracle chain', kinks=kinks)
    worst['_kinks'] = kinks
    return worst


def make_seq_state(res=80, body_sub=3, tube=(20, 6), seed=0, cloth_z=0.2):
    """a seeded oracle state of the seq stage at another size: make_state's body model / camera / material, a closed body ellipsoid + an open
    garment tube whose back cuts into the body (as the golden's scene), labels / connectivity prepared as train.py:1885-1911, the non-rigid
    network of seq.npz with a fresh pose code"""
    from d3h import synth
    from oracle import seq_ops as OS
    st = make_state(n=6, res=res, frames=1, seed=seed, n_samples=0, ssim_weight=0.0)
    gen = torch.Generator().manual_seed(3000 + seed)
    bv, bf = synth.icosphere(body_sub)
    cv, cf = synth.tube(*tube)
    body_v = torch.from_numpy(bv) * torch.tensor([0.5, 0.75, 0.42]) + torch.tensor([0.0, -0.35, 0.0]) + 0.004 * torch.randn(bv.shape, generator=gen)
    cloth_v = torch.from_numpy(cv) * torch.tensor([0.62, 0.22, 0.50]) + torch.tensor([0.0, -0.40, cloth_z]) + 0.004 * torch.randn(cv.shape, generator=gen)
    v = torch.cat([body_v, cloth_v]).contiguous()
    f = torch.cat([torch.from_numpy(bf), torch.from_numpy(cf) + body_v.shape[0]]).long().contiguous()
    face_labels = torch.cat([torch.zeros(bf.shape[0], dtype=torch.long), torch.ones(cf.shape[0], dtype=torch.long)])
    counts = torch.bincount(f.reshape(-1) * 2 + face_labels.unsqueeze(1).expand(-1, 3).reshape(-1), minlength=v.shape[0] * 2)
    v_labels = counts.reshape(v.shape[0], 2).argmax(dim=1)
    conn, edges = OS.find_connected_faces(f)
    sq = golden('seq.npz')
    leaf = lambda t: t.clone().requires_grad_(True)
    H, W = st['res']
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')

    def ell(cx, cy, rx, ry, albedo):
        msk = ((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) < 1).float()[None, ..., None]
        return torch.cat([torch.tensor(albedo).expand(1, H, W, 3) * msk, msk], -1).contiguous()
    st['cloth_img'] = ell(0.51 * W, 0.42 * H, 0.19 * W, 0.15 * H, [0.30, 0.50, 0.65])
    st['body_img'] = ell(0.50 * W, 0.48 * H, 0.16 * W, 0.27 * H, [0.60, 0.45, 0.35])
    st['flags'] = dict(lambda_kd=0.1, lambda_ks=0.05, lambda_nrm=0.025, lambda_chroma=0.05, grid_res=12)
    st['iteration'] = 3
    st['seq'] = {'base_v': v, 'base_f': f, 'cloth_v': v[v_labels == 1], 'body_v': v[v_labels == 0], 'v_labels': v_labels, 'face_labels': face_labels,
                 'connected_faces': conn, 'edges': edges, 'body_f': f[face_labels == 0], 'fix_code': leaf(0.1 * torch.randn(1, 1, 136, generator=gen)),
                 'nr_sd': {k[6:]: leaf(torch.from_numpy(sq[k])) for k in sq.files if k.startswith('nr_sd.')},
                 'skip_layers': [int(x) for x in sq['nr_skip_layers']]}
    return st


def check_tick_seq_vs_oracle(dev, loss_tol=2e-4, grad_tol=2e-3, **kw):
    """the product's tick_seq on its DEFAULT buffers (MSE + cosine normal term, fused MLP_deform, mesh_ops kernels) against the oracle chain
    (pinned by tick_seq.npz) on a seeded state of another size; raster decisions compared on their own, then shared"""
    from oracle import tick as OTK, raster as OR, render as ORD
    st = make_seq_state(**kw)
    H, W = st['res']
    P = build_product(dev, st, 12, None)
    torch.manual_seed(78)
    draws = [ORD.draw_jitter(1, H, W)]
    from render import render as R
    cap = {}
    orig = R.render_mesh

    def grab(*a, **k):
        k['_keep_rast'] = True
        o = orig(*a, **k)
        cap['rast'] = o['_rast']
        return o
    R.render_mesh = grab
    try:
        with fixed_render_draws(draws, dev):
            r = P['geometry'].tick_seq(P['glctx'], P['target'], None, P['material'], P['loss_fn'], st['iteration'], None, t='all')
    finally:
        R.render_mesh = orig
    rast_p = cap['rast'].detach().cpu()
    with torch.no_grad():
        mo = OTK.get_mesh_seq(st, 0)
        rast_own, _ = OR.rasterize(ORD.xfm_points(mo['posed'][None], st['mvp'][:1]), st['seq']['base_f'], H, W)
    id_diff = rast_p[..., 3] != rast_own[..., 3]
    assert int(id_diff.sum()) <= 3, f'{int(id_diff.sum())} pixels differ in triangle id from the oracle rasteriser'
    assert (rast_p[..., 2] - rast_own[..., 2])[~id_diff].abs().max() < 2e-6
    ro = OTK.tick_seq(st, draws=draws[0], keep=True, rast_zw=rast_p[..., 2], rast_ids=rast_p[..., 3])
    for k in SEQ_KEYS:
        a, b = float(r[k]), float(ro[k])
        assert abs(a - b) <= loss_tol * max(1e-6, abs(b)), (k, a, b)
    img_p, tot_p = seq_totals(r)
    assert abs(float(tot_p) - float(ro['total'])) <= loss_tol * abs(float(ro['total']))
    names = seq_params(P)
    kinks = relu_kinks(st, ro)
    o_params = [('nr.' + k, p) for k, p in st['seq']['nr_sd'].items()] + [('fix_code', st['seq']['fix_code']), ('trans', st['trans'])]
    gi = torch.autograd.grad(img_p, [p for _, p in names], retain_graph=True, allow_unused=True)
    go = torch.autograd.grad(ro['img_part'], [p for _, p in o_params], retain_graph=True, allow_unused=True)
    worst = _cmp_grads({k: x for (k, _), x in zip(names, gi)}, {k: x for (k, _), x in zip(o_params, go)}, grad_tol, 'tick_seq vs oracle, image part',
                       kinks=kinks)
    tot_p.backward()
    ro['total'].backward()
    got = {k: p.grad for k, p in names}
    got['table'] = P['tex'].encoder.params.grad
    ref = {k: p.grad for k, p in o_params}
    ref['table'] = st['material']['table'].grad
    worst.update({'total.' + k: v for k, v in _cmp_grads(got, ref, grad_tol, 'tick_seq vs oracle', kinks=kinks).items()})
    worst['_kinks'] = kinks
    return worst


def check_launch_ahead(dev, res=32, grid_n=6, frames=2, ticks=4, prefit=150, body_verts=300, samples=96, loss_set='mask'):
    """geometry/hmsdf.py:_extract with the speculative extraction: nearest vertex, LBS, the surface samples and the first eikonal sweep queued at the
    vertex buffer's capacity BEFORE the host knows the sizes (row count read on the device) -- every loss term and every gradient of the tick equal to
    the plain order (D3H_LAUNCH_AHEAD=0 / D3H_MTETS_SPECULATE=0), over ticks whose surface grows and shrinks; the ahead path was really taken"""
    from d3h.scene import Scene
    from d3h import mtets
    from geometry import hmsdf
    ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4
    torch.manual_seed(0)
    sc = Scene(res=res, grid_n=grid_n, n_frames=frames, device=dev, prefit_steps=prefit, loss_set=loss_set, body_verts=body_verts, sdf_fn=ell,
               flags_hook=lambda F: (setattr(F, 'prefit_with_library_path', True), setattr(F, 'eikonal_samples', samples)))
    bg = torch.rand(frames, res, res, 3, device=dev)
    g = sc.geometry
    cl = lambda t: None if t is None else t.clone()

    def tick(it):
        torch.manual_seed(10 + it)
        sc._zero_grad()
        r = g.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, it, None)
        (r['d3h_total'] if 'd3h_total' in r else (r['msk_loss'] + r['reg_loss'])).backward()
        if torch.cuda.is_available() and str(dev).startswith('cuda'):
            torch.cuda.synchronize()
        out = {k: float(v.detach()) for k, v in r.items() if torch.is_tensor(v) and v.numel() == 1}
        grads = [cl(g.deform.grad), cl(sc.FLAGS.trans_optim.grad)] + [cl(p.grad) for p in g.sdf_net.parameters()]
        return out, grads, int(g.last_mesh_dict['imesh'].v_pos.shape[-2])

    keep = (mtets.SPECULATE, hmsdf.AHEAD)
    try:
        results, taken = {}, []
        for mode in ('ahead', 'plain'):
            mtets.SPECULATE = hmsdf.AHEAD = (mode == 'ahead')
            mtets.TetGrid._cache.clear()
            s0 = dict(mtets.SPEC_STATS)
            res_ = []
            for it in range(ticks):
                with torch.no_grad():
                    g.deform.fill_(0.15 * (it % 3))          # moves the surface: the sizes change from tick to tick
                if mode == 'ahead' and it == ticks - 1:
                    # the last tick outgrows its capacities: the speculative kernels write nothing, what was launched ahead is dropped (the
                    # first eikonal sweep included) and the tick falls back to the plain order -- same results
                    grid = mtets.TetGrid.get(g.indices)
                    grid.caps = {k: (8, 8, 8) for k in grid.caps}
                res_.append(tick(it))
            results[mode] = res_
            taken.append(mtets.SPEC_STATS['speculated'] - s0['speculated'])
            if mode == 'ahead':
                assert mtets.SPEC_STATS['overflowed'] - s0['overflowed'] == 1
        assert taken[0] >= ticks - 1 and taken[1] == 0, taken
        assert len({r[2] for r in results['plain']}) > 1, 'the surface did not change between the ticks'
        for it, ((la, ga, na), (lb, gb, nb)) in enumerate(zip(results['ahead'], results['plain'])):
            assert na == nb
            for k in la:
                # (1e-4: ssim_loss = 1 - mean(SSIM) ~ 0.02 amplifies the 1e-7 order-of-summation noise of the mean fifty-fold: 1.2e-5 measured)
                assert abs(la[k] - lb[k]) <= 1e-4 * max(1e-6, abs(lb[k])), (it, k, la[k], lb[k])
            assert na > 0 and sum(x is not None for x in gb) >= 10
            for j, (a, b) in enumerate(zip(ga, gb)):          # float atomics: same addends, another order
                assert (a is None) == (b is None), (it, j)
                if a is not None:
                    assert (a - b).norm() <= 1e-4 * b.norm() + 1e-9, (it, j, float((a - b).norm()), float(b.norm()))
    finally:
        mtets.SPECULATE, hmsdf.AHEAD = keep
