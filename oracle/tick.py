"""Oracle for one training tick of each stage (init / split / seq): the whole chain on the CPU (torch autograd; TEST INFRASTRUCTURE).

Restates geometry/hmsdf.py:416-523 (getMesh_init: SDF sweep -> GShell_Tets -> SMPLX_Deformer.lbs_forward -> auto_normals), :706-737
(render_init) and :810-915 (tick_init: mask MSE x 100, log-sRGB L1 image loss + the two msdf_image terms, eikonal with the hard-coded
schedule, SDF sign-change regulariser with its ramp, the normal term) by composing the per-op oracles (oracle.sdf_mlp,
oracle.marching_tets, oracle.lbs, oracle.render, oracle.image_ops), and the total of train.py:718 (reg + normal + msk).
Pinned by tests/golden/tick_init.npz: the reference's own HmSDFTetsGeometry.tick_init run in the dev container on a miniature scene
(tools/gen_golden.py:gen_tick_init), every loss term and the gradients of the total.

`state` is a plain dict of CPU tensors (leaves that require grad receive .grad):
  verts [N,3], indices [T,4] int64, deform [N,3], msdf [N], max_disp (float), sd {net.i.weight/bias},
  body {model dict for oracle.lbs: v_template, J_regressor, shapedirs, expr_dirs, parents, weights}, tmpl [V,3], A0 [J,4,4],
  shape [1,S], expr [F,E], root_pose [F,3], body_pose [F,63], jaw_pose [F,3], trans [F,3], offsets (face/joint/locator or None),
  mvp [B,4,4], campos [B,3], res (H, W), material {table, w1, w2, w3, bbox, omin, omax},
  all_img [B,H,W,4], all_normal [B,H,W,3], background [B,H,W,3], sampled_pts [S,3] | None,
  iteration, n_iter (FLAGS.iter), sdf_regularizer, eikonal_scale (None = schedule), ssim_weight, loss_set, normal_loss_fn, frames.
Build-side extensions covered: several frames per batch, each posed with its own parameters (the reference poses with idx[0],
hmsdf.py:471); the MSE + 0.1 (1 - cos) normal term of hmsdf.py:1067-1068 when no perceptual network is given; the SSIM term of
BASELINE config 3.
"""
import torch
import torch.nn.functional as F

from . import fl as _fl          # (`fl` is a local name -- the flags dict -- in tick_split / tick_seq)
from . import image_ops as OI
from . import lbs as OL
from . import marching_tets as OMT
from . import render as ORD
from . import sdf_mlp as OMLP

BASE_EDGES = [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3]


def all_edges(indices):
    """hmsdf.py:382-387"""
    e = indices[:, torch.tensor(BASE_EDGES)].reshape(-1, 2)
    return torch.unique(torch.sort(e, dim=1)[0], dim=0)


def frame_transforms(st, frames):
    """SMPLX layer -> A per frame (body_models.py:1225-1257 + lbs.py:216-247,311-413 through oracle.lbs)"""
    As = []
    z3, z45 = torch.zeros(1, 3), torch.zeros(1, 45)
    for f in frames:
        J = OL.joints_from_shape(st['body'], st['shape'], st['expr'][f:f + 1], st.get('face_offset'), st.get('joint_offset'), st.get('locator_offset'))
        fp = OL.full_pose(st['root_pose'][f:f + 1], st['body_pose'][f:f + 1], st['jaw_pose'][f:f + 1], z3, z3, z45, z45)
        As.append(OL.pose_transforms(st['body'], fp, J)[0])
    return torch.stack(As)


def get_mesh_init(st, frames):
    v_def = st['verts'] + st['max_disp'] * st['deform']                          # hmsdf.py:433
    sdf = OMLP.mlp_forward(v_def, st['sd'])
    mt = OMT.gshell_tets(v_def, sdf, st['msdf'], st['indices'])
    verts, faces = mt['verts'], mt['faces']
    A = frame_transforms(st, frames)
    posed = []
    for k, f in enumerate(frames):
        o, _, _ = OL.lbs_forward(verts, st['tmpl'], st['body']['weights'], st['A0'], A[k], st['trans'][f])
        posed.append(o)
    return {'v_def': v_def, 'sdf': sdf, 'mt': mt, 'verts': verts, 'faces': faces, 'posed': torch.stack(posed), 'A': A}


def eikonal(st, pts, iteration):
    """hmsdf.py:856-876"""
    v = pts.detach().clone().requires_grad_(True)
    s = OMLP.mlp_forward(v, st['sd'])
    es = st.get('eikonal_scale')
    coeff = es if es is not None else (3e-1 if iteration < 500 else (1e-1 if iteration < 2000 else 1e-2))
    g = torch.autograd.grad(s.sum(), v, create_graph=True)[0]
    return coeff * (g.pow(2).sum(dim=-1).sqrt() - 1).pow(2).mean()


def tick_init(st, buffers=('shaded', 'geometric_normal', 'msdf_image'), draws=None, keep=False, rast_zw=None, rast_ids=None):
    frames = st.get('frames') or list(range(st['mvp'].shape[0]))
    it = st['iteration']
    m = get_mesh_init(st, frames)
    need_vn = buffers is None or bool(set(buffers) & {'normal', 'normal_grad'})
    vn_posed = torch.stack([OI.auto_normals(p, m['faces']) for p in m['posed']]) if need_vn else m['posed']
    stages = {} if keep else None
    b = ORD.render_mesh(m['posed'], m['verts'], m['faces'], vn_posed, st['mvp'], st['campos'], st['res'], st['material'],
                        background=st['background'], msdf=m['mt']['msdf'], draws=draws, buffers=buffers, keep=stages, rast_zw=rast_zw, rast_ids=rast_ids)
    color_ref = st['all_img']
    gt_mask = color_ref[..., 3:]
    out = {}
    out['msk_loss'] = 100 * F.mse_loss(b['shaded'][..., 3:], color_ref[..., 3:])                          # hmsdf.py:835
    img = OI.image_loss(b['shaded'][..., 0:3] * gt_mask, color_ref[..., 0:3] * gt_mask, 'l1', 'log_srgb')  # train.py:81 ('logl1')
    if 'msdf_image' in b:
        img = img + 5e-1 * F.l1_loss(b['msdf_image'].clamp(min=0) * _fl(gt_mask == 0), torch.zeros_like(gt_mask))
        img = img + 5e-1 * F.l1_loss(b['msdf_image'].clamp(max=0) * _fl(gt_mask == 1), torch.ones_like(gt_mask))
    out['img_loss'] = img
    pts = st.get('sampled_pts')
    out['eik_loss'] = eikonal(st, pts, it) if pts is not None else torch.zeros(())
    t_iter = it / st['n_iter']
    sdf_weight = st['sdf_regularizer'] - (st['sdf_regularizer'] - 0.01) * min(1.0, 4.0 * t_iter)             # hmsdf.py:881
    out['sdf_reg_loss'] = OI.sdf_reg_loss(m['sdf'], all_edges(st['indices'])) * sdf_weight
    out['reg_loss'] = out['geo_reg_loss'] = out['sdf_reg_loss'] + out['eik_loss']
    if 'geometric_normal' in b and st.get('all_normal') is not None:
        out_n = F.normalize(b['geometric_normal'][..., 0:3], p=2, dim=-1) * torch.tensor([1.0, -1.0, -1.0])   # hmsdf.py:895-897
        gt_n = F.normalize(st['all_normal'][..., 0:3], p=2, dim=-1)
        nfn = st.get('normal_loss_fn')
        if nfn is not None:                                                                                   # hmsdf.py:899-902
            out['normal_loss'] = 50 * nfn(((out_n + 1.0) / 2.0).permute(0, 3, 1, 2), ((gt_n + 1.0) / 2.0).permute(0, 3, 1, 2))
        else:                                                                                                 # hmsdf.py:1067-1068
            out['normal_loss'] = F.mse_loss(out_n, gt_n) + 0.1 * (1 - F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean())
    else:
        out['normal_loss'] = torch.zeros(())
    sw = st.get('ssim_weight', 0.0)
    if sw:
        a = (b['shaded'][..., 0:3] * gt_mask).permute(0, 3, 1, 2)
        c = (color_ref[..., 0:3] * gt_mask).permute(0, 3, 1, 2)
        out['ssim_loss'] = sw * (1.0 - OI.ssim(a, c))
    if st.get('loss_set', 'full') == 'mask':
        out['total'] = out['msk_loss']
    else:
        out['total'] = out['reg_loss'] + out['normal_loss'] + out['msk_loss'] + out.get('ssim_loss', 0.0)     # train.py:718
    if keep:
        out['_mesh'], out['_buffers'], out['_stages'] = m, b, stages
    return out


def material_smoothness_grad(kd_grad, ks_grad, nrm_grad, lambda_kd, lambda_ks, lambda_nrm):
    """render/regularizer.py:47-52"""
    kd_luma = (kd_grad[..., 0] + kd_grad[..., 1] + kd_grad[..., 2]) / 3
    return torch.mean(kd_luma * kd_grad[..., -1]) * lambda_kd + torch.mean(ks_grad[..., :-1] * ks_grad[..., -1:]) * lambda_ks + \
        torch.mean(nrm_grad[..., :-1] * nrm_grad[..., -1:]) * lambda_nrm


def chroma_loss(kd, color_ref, lambda_chroma):
    """render/regularizer.py:19-26"""
    value = lambda x: torch.max(x[..., 0:3], dim=-1, keepdim=True)[0].repeat(1, 1, 1, 3)
    ref = color_ref[..., 0:3] / torch.clip(value(color_ref), min=0.001)
    opt = kd[..., 0:3] / torch.clip(value(kd), min=0.001)
    return torch.mean(torch.abs((opt - ref) * color_ref[..., 3:])) * lambda_chroma


def crop_pair(a, b, h, w, crop_size, rng):
    """hmsdf.py:68-76 (`rng`: a random.Random or the `random` module; width offset is drawn first)"""
    sw = rng.randint(0, w - crop_size)
    sh = rng.randint(0, h - crop_size)
    return a[..., sh:sh + crop_size, sw:sw + crop_size], b[..., sh:sh + crop_size, sw:sw + crop_size]


def get_mesh_split(st, frames, type):
    """hmsdf.py:526-630: the sweep of getMesh_init, then hmSDF_Tets (type == "body": the mSDF negated inside no_grad)"""
    v_def = st['verts'] + st['max_disp'] * st['deform']
    sdf = OMLP.mlp_forward(v_def, st['sd'])
    mt = OMT.gshell_tets(v_def, sdf, st['msdf'], st['indices'], negate_msdf=(type == 'body'))
    verts, faces = mt['verts'], mt['faces']
    A = frame_transforms(st, frames)
    posed = []
    for k, f in enumerate(frames):
        o, _, _ = OL.lbs_forward(verts, st['tmpl'], st['body']['weights'], st['A0'], A[k], st['trans'][f])
        posed.append(o)
    return {'v_def': v_def, 'sdf': sdf, 'mt': mt, 'verts': verts, 'faces': faces, 'posed': torch.stack(posed), 'A': A}


def tick_split(st, type, draws=None, pts=None, rng=None, keep=False, rast_zw=None, rast_ids=None):
    """hmsdf.py:917-1096 for type in {"cloth", "body", "all"}.  st carries, beyond tick_init's keys: '<type>_img', '<type>_normal',
    'flags' = dict(use_mesh_msdf_reg, msdf_reg_open_scale, msdf_reg_close_scale, lambda_kd, lambda_ks, lambda_nrm, lambda_chroma,
    texture_res, grid_res); 'lpips_fn' (build-side extension: added to img_loss).  `pts`: the surface samples of THIS call."""
    import random as _random
    frames = st.get('frames') or list(range(st['mvp'].shape[0]))
    it = st['iteration']
    fl = st['flags']
    m = get_mesh_split(st, frames, type)
    vn_posed = torch.stack([OI.auto_normals(p, m['faces']) for p in m['posed']])
    stages = {} if keep else None
    b = ORD.render_mesh(m['posed'], m['verts'], m['faces'], vn_posed, st['mvp'], st['campos'], st['res'], st['material'],
                        background=st['background'], msdf=m['mt']['msdf'], draws=draws, buffers=None, keep=stages, rast_zw=rast_zw, rast_ids=rast_ids)
    color_ref, normal_ref = st[type + '_img'], st[type + '_normal']
    gt_mask = color_ref[..., 3:]
    out = {}
    out['msk_loss'] = F.mse_loss(b['shaded'][..., 3:], color_ref[..., 3:])                                 # hmsdf.py:947 (no factor 100)
    img = OI.image_loss(b['shaded'][..., 0:3] * gt_mask, color_ref[..., 0:3] * gt_mask, 'l1', 'log_srgb')
    img = img + 5e-1 * F.l1_loss(b['msdf_image'].clamp(min=0) * _fl(gt_mask == 0), torch.zeros_like(gt_mask))
    img = img + 5e-1 * F.l1_loss(b['msdf_image'].clamp(max=0) * _fl(gt_mask == 1), torch.ones_like(gt_mask))
    lp = st.get('lpips_fn')
    if lp is not None:
        out['lpips_loss'] = lp((b['shaded'][..., 0:3] * gt_mask).permute(0, 3, 1, 2), (color_ref[..., 0:3] * gt_mask).permute(0, 3, 1, 2)).mean() * \
            st.get('lpips_weight', 1.0)
        img = img + out['lpips_loss']
    out['img_loss'] = img
    out['eik_loss'] = eikonal(st, pts, it) if pts is not None else torch.zeros(())
    if fl.get('use_mesh_msdf_reg', True):                                                                # hmsdf.py:996-1028
        regscale = (64 / fl['grid_res']) ** 3
        eps = torch.tensor([1e-3])
        reg = torch.zeros(())
        if fl['msdf_reg_open_scale'] > 0:
            mm = m['mt']['msdf']
            reg = fl['msdf_reg_open_scale'] * regscale * F.huber_loss(mm.clamp(min=-eps).reshape(-1), -eps.expand(mm.shape[0]), reduction='sum')
        if fl['msdf_reg_close_scale'] != 0:
            with torch.no_grad():
                n_wt = m['mt']['n_verts_watertight']
                vv = m['faces'][b['visible_triangles']].unique()
                vb = vv[vv >= n_wt] - n_wt
                mask = torch.zeros(m['mt']['msdf_boundary'].shape[0], dtype=torch.bool)
                mask[vb] = True
            bm = m['mt']['msdf_boundary'][mask]
            reg = reg + fl['msdf_reg_close_scale'] * regscale * F.huber_loss(bm.clamp(max=eps).reshape(-1), eps.expand(bm.shape[0]), reduction='sum')
        out['mesh_msdf_reg_loss'] = reg
    else:
        out['mesh_msdf_reg_loss'] = torch.zeros(())
    t_iter = it / st['n_iter']
    sdf_weight = st['sdf_regularizer'] - (st['sdf_regularizer'] - 0.01) * min(1.0, 4.0 * t_iter)
    out['sdf_reg_loss'] = OI.sdf_reg_loss(m['sdf'], all_edges(st['indices'])) * sdf_weight
    out['monochrome_loss'] = torch.zeros_like(img)
    out['mtl_smooth_loss'] = material_smoothness_grad(b['kd_grad'], b['ks_grad'], b['normal_grad'], fl['lambda_kd'], fl['lambda_ks'], fl['lambda_nrm'])
    out['chroma_loss'] = chroma_loss(b['kd'], color_ref, fl['lambda_chroma'])
    out['geo_reg_loss'] = out['sdf_reg_loss'] + out['eik_loss']
    out['shading_reg_loss'] = out['monochrome_loss'] + out['mtl_smooth_loss'] + out['chroma_loss']
    out['reg_loss'] = out['geo_reg_loss'] + out['shading_reg_loss']
    out_n = F.normalize(b['geometric_normal'][..., 0:3], p=2, dim=-1) * torch.tensor([1.0, -1.0, -1.0])
    gt_n = F.normalize(normal_ref[..., 0:3], p=2, dim=-1)
    out['normal_loss_mse'] = F.mse_loss(out_n, gt_n)                                                       # hmsdf.py:1067-1068
    out['normal_loss_cos'] = 0.1 * (1 - F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean())
    nfn = st.get('normal_loss_fn')
    if nfn is not None:                                                                                   # hmsdf.py:1069-1074
        a, c = ((out_n + 1.0) / 2.0).permute(0, 3, 1, 2), ((gt_n + 1.0) / 2.0).permute(0, 3, 1, 2)
        a, c = crop_pair(a, c, fl['texture_res'][0], fl['texture_res'][1], 448, rng if rng is not None else _random)
        out['normal_loss'] = 5 * nfn(a, c)
    else:
        out['normal_loss'] = out['normal_loss_mse'] + out['normal_loss_cos']
    out['depth_loss'] = out['delta_loss'] = torch.zeros(())            # use_depth False; iteration <= nonrigid_begin (hmsdf.py:961-968,1055-1061)
    out['total'] = out['img_loss'] + out['normal_loss'] + out['reg_loss'] + 10 * out['msk_loss']            # train.py:1050,1067,1087 (one half)
    if keep:
        out['_mesh'], out['_buffers'], out['_stages'] = m, b, stages
    return out


def get_mesh_seq(st, frame):
    """hmsdf.py:632-704 with a target: the non-rigid network on cloth_v / body_v scattered by v_labels, LBS of base_v + delta"""
    from . import seq_ops as OS
    sq = st['seq']
    delta = torch.zeros_like(sq['base_v'])
    fwd = lambda x: OS.mlp_deform_forward(x.reshape(1, -1, 3), sq['fix_code'], sq['nr_sd'], n_freq=8, skip_layers=tuple(sq['skip_layers'])).reshape(-1, 3)
    delta = delta.index_put((torch.nonzero(sq['v_labels'] == 1).reshape(-1),), fwd(sq['cloth_v']))
    delta = delta.index_put((torch.nonzero(sq['v_labels'] == 0).reshape(-1),), fwd(sq['body_v']))
    delta_v = sq['base_v'] + delta
    A = frame_transforms(st, [frame])
    posed, _, _ = OL.lbs_forward(delta_v, st['tmpl'], st['body']['weights'], st['A0'], A[0], st['trans'][frame])
    return {'delta': delta, 'delta_v': delta_v, 'posed': posed}


def tick_seq(st, draws=None, keep=False, rast_zw=None, rast_ids=None):
    """hmsdf.py:1099-1182 (render_seq :776-808, render_mask.render_mesh) + the total of train.py:1412-1421.  st['seq'] = dict(base_v,
    base_f, cloth_v, body_v, v_labels, face_labels, connected_faces, edges, body_f, fix_code, nr_sd, skip_layers); targets cloth_img,
    body_img, all_img, all_normal; flags lambda_*."""
    from . import seq_ops as OS
    sq, fl = st['seq'], st['flags']
    frame = (st.get('frames') or [0])[0]
    m = get_mesh_seq(st, frame)
    f = sq['base_f']
    v = m['posed']
    vn = OI.auto_normals(v, f)
    stages = {} if keep else None
    b = ORD.render_mesh(v[None], sq['base_v'], f, vn[None], st['mvp'][:1], st['campos'][:1], st['res'], st['material'], background=st['background'][:1],
                        msdf=None, draws=draws, buffers=None, keep=stages, rast_zw=rast_zw, rast_ids=rast_ids, face_labels=sq['face_labels'])
    lab = b['mesh_id'][..., 0]
    alpha = b['geometric_normal'][..., -1]
    m_cloth, m_body, m_all = (lab * alpha)[..., None], ((1 - lab) * alpha)[..., None], alpha[..., None]      # hmsdf.py:790-797
    gt_all, gt_cloth, gt_body = st['all_img'], st['cloth_img'], st['body_img']
    loss = lambda a, c: OI.image_loss(a, c, 'l1', 'log_srgb')
    out = {'delta': m['delta']}
    out['all_msk_loss'] = 200 * F.mse_loss(m_all, gt_all[..., 3:])
    out['cloth_msk_loss'] = 200 * F.mse_loss(m_cloth, gt_cloth[..., 3:])
    out['body_msk_loss'] = 200 * F.mse_loss(m_body, gt_body[..., 3:])
    rgb = b['shaded'][..., 0:3]
    out['all_img_loss'] = loss(rgb * m_all, gt_all[..., 0:3])
    out['cloth_img_loss'] = loss(rgb * m_cloth, gt_cloth[..., 0:3])
    out['body_img_loss'] = loss(rgb * m_body, gt_body[..., 0:3])
    out['mtl_smooth_loss'] = material_smoothness_grad(b['kd_grad'], b['ks_grad'], b['normal_grad'], fl['lambda_kd'], fl['lambda_ks'], fl['lambda_nrm'])
    out['chroma_loss'] = chroma_loss(b['kd'], gt_all, fl['lambda_chroma'])
    out['shading_reg_loss'] = out['reg_loss'] = out['mtl_smooth_loss'] + out['chroma_loss']
    out['delta_loss'] = torch.sum(torch.norm(m['delta'], dim=1) ** 2)
    out_n = F.normalize(b['geometric_normal'][..., 0:3], p=2, dim=-1) * torch.tensor([1.0, -1.0, -1.0])
    gt_n = F.normalize(st['all_normal'][..., 0:3], p=2, dim=-1)
    nfn = st.get('normal_loss_fn')
    if nfn is not None:                                                                                   # hmsdf.py:1150-1154
        out['normal_loss'] = 20 * nfn(((out_n + 1.0) / 2.0).permute(0, 3, 1, 2), ((gt_n + 1.0) / 2.0).permute(0, 3, 1, 2))
    else:
        out['normal_loss'] = F.mse_loss(out_n, gt_n) + 0.1 * (1 - F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean())
    # Mesh.__init__ replaces the `edges` it is handed (FLAGS.edges: all 3F sorted edges, train.py:1900) with the unique edges of its own
    # faces (mesh.py:158-162,240-250): on an open mesh the two give different vertex degrees, and the unique ones are what the loss sees
    out['laplacian_loss'] = OS.laplacian_uniform_loss(v, OS.find_edges(f))
    out['nds_normal_loss'] = OS.normal_consistency_loss(v, f, sq['connected_faces'])
    out['colli_loss'] = OS.collision_loss(v[sq['v_labels'] == 1], v[sq['v_labels'] == 0], sq['body_f'])
    out['visible_triangles'] = b['visible_triangles']
    out['img_part'] = 250 * out['normal_loss'] + 0.1 * out['reg_loss'] + (out['body_msk_loss'] + out['cloth_msk_loss'] + out['all_msk_loss'])
    out['total'] = out['img_part'] + \
        1000000 * out['laplacian_loss'] + 100000 * out['colli_loss'] + 1000 * out['nds_normal_loss'] + out['delta_loss']      # train.py:1412-1421
    if keep:
        out['_mesh'], out['_buffers'], out['_stages'] = m, b, stages
    return out


def surface_samples(verts, faces, n, generator=None):
    """kaolin.ops.mesh.sample_points (un-vendored): face ~ Multinomial(area), (u, v) ~ U^2,
    p = (1 - sqrt(u)) a + sqrt(u) (1 - v) b + sqrt(u) v c"""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * torch.cross(b - a, c - a, dim=-1).norm(dim=-1)
    fi = torch.multinomial(area, n, replacement=True, generator=generator)
    uv = torch.rand(n, 2, generator=generator)
    su = uv[:, 0:1].sqrt()
    return (1 - su) * a[fi] + su * (1 - uv[:, 1:2]) * b[fi] + su * uv[:, 1:2] * c[fi]


def state_from_golden(g, perceptual_cls=None):
    """tests/golden/tick_init.npz (dict of numpy arrays) -> the `state` of this module, with fresh leaves; `perceptual_cls` builds the
    MobileNetV2-shaped trunk of the normal loss from the golden's seed (oracle.perceptual.MobileNetPerceptualLoss)"""
    import numpy as np
    from . import texmlp as OT
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))
    leaf = lambda t: t.clone().requires_grad_(True)
    gen = torch.Generator().manual_seed(int(g['enc_seed']))
    table = (torch.rand(2 * OT.grid_layout()[1], generator=gen) * 2 - 1) * float(g['enc_scale'])
    body = {k[6:]: T(k) for k in g if k.startswith('model.')}
    st = {'verts': T('verts'), 'indices': T('indices'), 'deform': leaf(T('deform')), 'msdf': leaf(T('msdf')),
          'max_disp': 1.0 / int(g['grid_res']) * 1.0 / 2.1, 'sd': {k[3:]: leaf(T(k)) for k in g if k.startswith('sd.')},
          'body': body, 'tmpl': T('tmpl'), 'A0': T('A0'), 'shape': T('betas'), 'expr': T('expr'), 'root_pose': T('root_pose'),
          'body_pose': T('body_pose'), 'jaw_pose': T('jaw'), 'trans': leaf(T('trans')), 'mvp': T('mvp'), 'campos': T('campos'),
          'res': (int(g['res']), int(g['res'])),
          'material': {'table': leaf(table), 'w1': leaf(T('w1')), 'w2': leaf(T('w2')), 'w3': leaf(T('w3')),
                       'bbox': (0.6, 0.6, 0.2, -0.8, -1.2, -0.2), 'omin': g['omin'].tolist(), 'omax': g['omax'].tolist()},
          'all_img': T('all_img'), 'all_normal': T('all_normal'), 'background': T('bg'), 'sampled_pts': T('sampled_pts') if 'sampled_pts' in g else None,
          'iteration': int(g['iteration']), 'n_iter': int(g['n_iter']), 'sdf_regularizer': float(g['sdf_regularizer']),
          'eikonal_scale': None, 'ssim_weight': 0.0, 'loss_set': 'full'}
    if 'sampled_pts' not in g:
        st['sampled_pts'] = None
    for k in ('cloth_img', 'cloth_normal', 'body_img', 'body_normal', 'sampled_pts.cloth', 'sampled_pts.body'):      # tick_split.npz / tick_seq.npz
        if k in g:
            st[k] = T(k)
    fl = {k[5:]: (g[k].tolist() if np.ndim(g[k]) else g[k].item()) for k in g if k.startswith('flag.')}
    if fl:
        fl['grid_res'] = int(g['grid_res'])
        st['flags'] = fl
    if 'seq.base_v' in g:
        sq = {k[4:]: T(k) for k in g if k.startswith('seq.') and not k.startswith('seq.nr_sd.')}
        sq['nr_sd'] = {k[10:]: leaf(T(k)) for k in g if k.startswith('seq.nr_sd.')}
        sq['fix_code'] = leaf(sq['fix_code'])
        sq['skip_layers'] = [int(x) for x in g['seq.skip_layers']]
        st['seq'] = sq
    if perceptual_cls is not None:
        st['normal_loss_fn'] = perceptual_cls(use_gpu=False, seed=int(g['trunk_seed']))
    return st
