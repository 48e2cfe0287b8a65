"""Oracle for the deferred renderer: render_mesh / render_layer / shade (torch CPU; TEST INFRASTRUCTURE).

Restates render/render.py:347-451 (render_mesh: clip transform, first-layer rasterize, visible_triangles, per-buffer composite +
antialias), :213-345 (render_layer: the six interpolations, face normals :261-267, the random tangent of the use_uv == False branch
:284-287, z / z-gradient :291-299, msdf_image :328) and :42-205 (shade with `bsdf = 'kd'` forced at :120 and perturbed_nrm None) on
top of oracle.raster (nvdiffrast restatement, parity unpinned there) and oracle.texmlp.  Pinned by tests/golden/render.npz: the
outputs of the reference's own render.py driven by the same oracle dr / tcnn (tools/gen_golden.py:gen_render), all 12 buffers.

Two build-side extensions are restated as well, so that the oracle chain can check them: `v_pos` may be [B,P,3] (one posed mesh per
frame of the batch) and the random draws can be handed in (`draws`), in the order the reference consumes the generator:
'noise' (randn_like(gb_normal), :285), 'offset' (normal(0, 0.005) [B,H,W,2], render.py:68), 'pos_noise' (normal(0, 0.01)
gb_pos.shape, :84).
"""
import torch
import torch.nn.functional as F

from . import fl
from . import raster as OR
from . import texmlp as OT
from . import image_ops as OI

ALL_BUFFERS = ('shaded', 'z_grad', 'normal', 'geometric_normal', 'kd', 'ks', 'kd_grad', 'ks_grad', 'normal_grad', 'depth', 'invdepth')


def xfm_points(points, matrix):
    """renderutils/ops.py:531"""
    return torch.matmul(F.pad(points, (0, 1), value=1.0), matrix.transpose(1, 2))


def pixel_grid(W, H):
    """render/util.py:61-65"""
    y, x = torch.meshgrid((fl(torch.arange(0, H)) + 0.5) / H, (fl(torch.arange(0, W)) + 0.5) / W, indexing='ij')
    return torch.stack((x, y), dim=-1)


def draw_jitter(B, H, W, generator=None):
    """the three random tensors of one render_mesh call, in the reference's order of consumption"""
    noise = torch.randn(B, H, W, 3, generator=generator)
    offset = torch.normal(mean=0, std=0.005, size=(B, H, W, 2), generator=generator)
    pos_noise = torch.normal(mean=0, std=0.01, size=(B, H, W, 3), generator=generator)
    return {'noise': noise, 'offset': offset, 'pos_noise': pos_noise}


def sample_material(mat, x):
    """render/mlptexture.py:91-107 (MLPTexture3D.sample): mat = dict(table, w1, w2, w3, bbox, omin, omax)"""
    return OT.texture_mlp(x, mat['table'], mat['w1'], mat['w2'], mat['w3'], mat['bbox'], mat['omin'], mat['omax'])


def render_mesh(v_pos, v_pos_orig, faces, v_nrm, mtx, view_pos, res, material, background=None, msdf=None, draws=None, buffers=None,
                antialias=True, keep=None, rast_zw=None, rast_ids=None, face_labels=None):
    """-> dict of [B,H,W,C+1] buffers (msdf_image: [B,H,W,1]) + 'visible_triangles' + '_rast'.
    v_pos [P,3] | [B,P,3]; v_pos_orig [P,3]; v_nrm like v_pos; faces int64 [F,3]; mtx [B,4,4]; view_pos [B,3]"""
    H, W = int(res[0]), int(res[1])
    want = set(ALL_BUFFERS) | {'msdf_image'} if buffers is None else set(buffers)
    vb = v_pos if v_pos.dim() == 3 else v_pos[None]
    nb = v_nrm if (v_nrm is None or v_nrm.dim() == 3) else v_nrm[None]
    clip = xfm_points(vb, mtx)
    B = clip.shape[0]
    vpos = view_pos[:, None, None, :] if view_pos.dim() == 2 else view_pos
    rast, db = OR.rasterize(clip, faces, H, W, ids=rast_ids)
    if rast_zw is not None:
        # z/w (and, with rast_ids, the per-pixel winners) of another rasteriser; the caller compares both with this oracle's own.  antialias decides which of two neighbouring pixels is
        # the nearer one by comparing their z/w; with the reference's near / far planes (0.001 / 1000, dataset_split.py:57-68) a whole
        # body spans ~10 ulps of z/w, so that comparison is decided by the last bit.  Handing both implementations the same z/w makes
        # their antialias passes comparable pixel by pixel; the z/w values themselves are compared separately.
        rast = torch.cat([rast[..., 0:2], rast_zw.reshape(rast.shape[:3])[..., None], rast[..., 3:4]], dim=-1)
    out = {'_rast': rast}
    vis = rast[..., -1].long().unique()                                        # render.py:404-407
    if vis.numel() and vis[0] == 0:
        vis = vis[1:]
    out['visible_triangles'] = vis - 1

    # ---- render_layer ------------------------------------------------------------------------------------------------------------
    gb_pos, _ = OR.interpolate(vb, rast, faces)
    gb_pos_orig, _ = OR.interpolate(v_pos_orig[None], rast, faces)
    v0, v1, v2 = vb[:, faces[:, 0]], vb[:, faces[:, 1]], vb[:, faces[:, 2]]
    fn = OI.safe_normalize(torch.cross(v1 - v0, v2 - v0, dim=-1))               # [B,F,3]
    ids = rast[..., 3].long()
    cov = (ids > 0)
    bi = torch.arange(B)[:, None, None].expand(-1, H, W) if fn.shape[0] > 1 else torch.zeros(B, H, W, dtype=torch.long)
    gb_gn = torch.where(cov[..., None], fn[bi, (ids - 1).clamp(min=0)], torch.zeros(B, H, W, 3))     # (f,f,f)-indexed interpolation
    gb_normal, _ = OR.interpolate(nb, rast, faces)
    if draws is None:
        draws = draw_jitter(B, H, W)
    noise = draws['noise']
    noise = noise / noise.norm(dim=-1, keepdim=True)
    gb_tangent = torch.cross(noise, gb_normal, dim=-1)
    with torch.no_grad():
        eps = 0.00001
        cp, cpd = OR.interpolate(clip, rast, faces, rast_db=db)
        z0 = torch.clamp(cp[..., 2:3], min=eps) / torch.clamp(cp[..., 3:4], min=eps)
        z1 = torch.clamp(cp[..., 2:3] + torch.abs(cpd[..., 2:3]), min=eps) / torch.clamp(cp[..., 3:4] + torch.abs(cpd[..., 3:4]), min=eps)
        gb_depth = torch.cat((z0, torch.abs(z1 - z0)), dim=-1)

    # ---- shade (bsdf == 'kd') -----------------------------------------------------------------------------------------------------
    jitter = (pixel_grid(W, H)[None] + draws['offset']).contiguous()
    mask = fl(rast[..., -1:] > 0)
    mask_tap = OR.texture(mask, jitter)
    grad_weight = mask * mask_tap
    layer = {}
    need_tex = want & {'shaded', 'kd', 'ks', 'kd_grad', 'ks_grad'}
    if need_tex:
        all_tex = sample_material(material, gb_pos_orig)
        kd, ks = all_tex[..., 0:3], all_tex[..., 3:6]
        if want & {'kd_grad', 'ks_grad'}:
            all_tex_jitter = sample_material(material, gb_pos_orig + draws['pos_noise'])
            layer['kd_grad'] = torch.abs(all_tex_jitter[..., 0:3] - kd)
            layer['ks_grad'] = torch.abs(all_tex_jitter[..., 3:6] - ks) * torch.tensor([0.0, 1.0, 1.0])
        layer['shaded'], layer['kd'], layer['ks'] = kd, kd, ks
    if 'normal_grad' in want:
        nrm_jitter = OR.texture(gb_normal, jitter)
        layer['normal_grad'] = torch.abs(nrm_jitter - gb_normal) * grad_weight
    if 'normal' in want:
        # perturbed_nrm is None: the kernel substitutes (0, 0, 1) (normal.cu / bsdf.py:25-51)
        pert = torch.tensor([0.0, 0.0, 1.0]).expand_as(gb_normal)
        layer['normal'] = OI.prepare_shading_normal(gb_pos, vpos, pert, gb_normal, gb_tangent, gb_gn, True, True)
    layer['z_grad'] = torch.cat((gb_depth, torch.zeros_like(gb_depth[..., 0:1])), dim=-1)
    layer['geometric_normal'] = gb_gn
    d = gb_pos - vpos
    layer['depth'] = d.pow(2).sum(dim=-1, keepdim=True).sqrt()
    layer['invdepth'] = 1.0 / (d.pow(2) + 1e-8).sum(dim=-1, keepdim=True).sqrt()

    # ---- composite + antialias (render.py:375-382,430-449) -----------------------------------------------------------------------
    covf = fl(cov[..., None])
    comp = {}
    for k in ALL_BUFFERS:
        if k not in want or k not in layer:
            continue
        val = torch.cat((layer[k], torch.ones(B, H, W, 1)), dim=-1)
        if k == 'shaded':
            bgk = torch.cat((background, torch.zeros_like(background[..., 0:1])), dim=-1) if background is not None else torch.zeros(1, H, W, 4)
        elif k == 'depth':
            bgk = torch.ones_like(val) * 20.0
        else:
            bgk = torch.zeros_like(val)
        comp[k] = torch.lerp(bgk.expand_as(val), val, covf)                      # alpha = coverage * 1
    if msdf is not None and 'msdf_image' in want:
        mi, _ = OR.interpolate(msdf.reshape(1, -1, 1), rast, faces)
        comp['msdf_image'] = torch.lerp(torch.zeros_like(mi), torch.ones_like(mi), covf * mi)
    if antialias and comp:
        # antialias is linear per channel and its discrete analysis depends on (rast, clip, faces) only: one pass over the
        # channel-concatenated image equals the reference's one pass per buffer
        keys = list(comp)
        pre = torch.cat([comp[k] for k in keys], dim=-1).contiguous()
        st = OR.antialias(pre, rast, clip, faces)
        if keep is not None:          # intermediates for stage-by-stage debugging of a product / oracle mismatch
            keep.update({'pre_aa': pre, 'post_aa': st, 'clip': clip, 'gb_pos_orig': gb_pos_orig, 'gb_gn': gb_gn, 'keys': keys, 'rast': rast})
        c0 = 0
        for k in keys:
            n = comp[k].shape[-1]
            out[k] = st[..., c0:c0 + n]
            c0 += n
    else:
        out.update(comp)
    if face_labels is not None:
        # render_mask.py:313,391-396,462-463: the covering triangle's label through an (f, f, f)-indexed interpolation (= a gather),
        # composited WITHOUT antialiasing: [label, 1] where covered and label != 0, [0, 0] elsewhere
        lab = torch.where(cov, fl(face_labels)[(ids - 1).clamp(min=0)], torch.zeros(B, H, W))[..., None]
        on = fl(cov[..., None] & (lab != 0))
        out['mesh_id'] = torch.cat((lab, torch.ones_like(lab)), dim=-1) * on
    return out
