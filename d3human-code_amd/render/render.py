"""Deferred renderer with the reference's entry point `render_mesh(...)` (render/render.py:347-451) on the MI355X kernels.

What the reference does per call (render.py:213-345,42-205,375-449): clip transform -> first-layer rasterize -> six separate
dr.interpolate calls -> shade (two texture-MLP sweeps over every pixel incl. background, prepare_shading_normal, bsdf forced to
'kd') -> for each of 12 buffers a lerp against its background and its own dr.antialias pass.  Same outputs here, regrouped for
HBM: attributes sharing an index buffer are interpolated in ONE pass, the texture MLP runs only on covered pixels, and all
buffers are composited and antialiased in ONE pass over a channel-concatenated image (antialias is per-channel linear, so this is
exact).  `buffers=` (an extension; default = all, as the reference) lets a caller name the outputs it will read.

`mesh.v_pos` may be [P,3] (the reference) or [B,P,3] (one posed mesh per frame of the batch: the build's N-frame extension,
SURVEY F5).  spp > 1 is not part of the hot path (FLAGS.spp = 1) and raises.
"""
import os

import torch
from d3h.devconst import const as _const
import nvdiffrast.torch as dr

from . import util
from . import renderutils as ru
from d3h import imgops as _I
from d3h import raster as _R

ALL_BUFFERS = ('shaded', 'z_grad', 'normal', 'geometric_normal', 'kd', 'ks', 'kd_grad', 'ks_grad', 'normal_grad', 'depth', 'invdepth')


class LazyVisibleTriangles(torch.Tensor):
    """The sorted ids of the triangles that own at least one pixel (render.py:404-407) as a DEFERRED tensor: the compaction
    `nonzero(bitmap)` has a data-dependent size, i.e. a host synchronisation in the middle of the iteration.  Internal to tick_seq, which
    hands `visible_triangles` back to a loop that reads it once, after the last iteration (train.py:1515); `render_mesh` itself returns a
    plain tensor whenever `visible_triangles` is asked for.
    A real torch.Tensor subclass: isinstance / torch.is_tensor hold, and EVERY use -- a method or property, an argument of a torch
    function at any nesting depth (`torch.cat([v, t])`), an index (`faces[v]`), copy / deepcopy / pickle -- goes through
    __torch_function__ (or __reduce_ex__), which swaps in the materialised int64 tensor first."""

    @staticmethod
    def __new__(cls, seen):
        return torch.Tensor._make_subclass(cls, torch.empty(0, dtype=torch.long, device=seen.device))

    def __init__(self, seen):
        self._d3h_seen, self._d3h_value = seen, None

    def materialize(self):
        if self._d3h_value is None:
            with torch._C.DisableTorchFunctionSubclass():
                self._d3h_value = torch.nonzero(self._d3h_seen).reshape(-1)
        return self._d3h_value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda a: a.materialize() if isinstance(a, LazyVisibleTriangles) else a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(un, tuple(args)), **tree_map(un, dict(kwargs or {})))

    def __reduce_ex__(self, proto):
        return self.materialize().__reduce_ex__(proto)

    def __repr__(self):
        return 'LazyVisibleTriangles(%s)' % ('<deferred>' if self._d3h_value is None else repr(self._d3h_value))


def interpolate(attr, rast, attr_idx, rast_db=None):
    return dr.interpolate(attr.contiguous(), rast, attr_idx, rast_db=rast_db, diff_attrs=None if rast_db is None else 'all')


def _batched(v):
    return v if v.dim() == 3 else v[None]


def shade(FLAGS, idx, rast, aux, gb_pos, gb_pos_original, gb_geometric_normal, gb_normal, gb_tangent, view_pos, material, want,
          finetune_normal=True, mask=None, rng_draws=None, live=None, skip_uncovered=True):
    """render.py:42-205 restricted to the live branch (bsdf == 'kd', perturbed_nrm is None).  `aux` = (z_grad values, depth, invdepth) from
    the fused forward-only pass (d3h.raster.aux_buffers; entries None where not produced); `live`: the buffers that need a gradient
    (None = all) -- the producers of the others run under torch.no_grad()."""
    B, H, W = rast.shape[:3]
    dev = rast.device
    grad_on = torch.is_grad_enabled()
    on = lambda *ks: torch.set_grad_enabled(grad_on and (live is None or any(k in live for k in ks)))
    if mask is None:
        mask = (rast[..., -1:] > 0).float()                                       # render.py:66
    need_jitter = bool(want & {'normal_grad', 'kd_grad', 'ks_grad'})
    # RNG call order follows the reference (offset, then the position jitter) so a seeded CPU run reproduces it
    if rng_draws is not None:            # pre-drawn jitter (tests: the same draws on every device)
        offset, pos_noise = rng_draws['offset'].to(dev), rng_draws['pos_noise'].to(dev)
    else:
        offset = torch.normal(mean=0, std=0.005, size=(B, H, W, 2), device=dev) if need_jitter else None
        pos_noise = torch.normal(mean=0, std=0.01, size=gb_pos_original.shape, device=dev) if need_jitter else None

    kd_ks = material['kd_ks']
    # the texture MLP skips uncovered pixels (their value never reaches an output: alpha = 0) -- except under supersampling, where a
    # covered sub-pixel can inherit the value of an uncovered shading pixel (render.py:241-245,334-336): then every pixel is evaluated
    tex_mask = mask if skip_uncovered else None
    out = {}
    tex_users = ('shaded', 'kd', 'ks', 'kd_grad', 'ks_grad')
    is_live = lambda k: live is None or k in live
    # kd and the three smoothness buffers in ONE pass each way (d3h.imgops.material_grads) when every one of them that is wanted takes a
    # gradient (a tick of the split / seq stage) or none does; in the mixed case (the 'all' mode of a tick) the dead ones stay separate
    # torch ops under no_grad so that the backward does not pay for them
    smooth = want & {'kd_grad', 'ks_grad', 'normal_grad'}
    fused = ({'kd_grad', 'ks_grad'} <= want) and (not grad_on or all(is_live(k) for k in smooth | (want & {'shaded', 'kd'})))
    # ... and when NONE of the smoothness buffers is live (tick_init in its 'all' mode) they come out of the same pass under no_grad, kd staying
    # the plain slice that carries the gradient
    fused_dead = (not fused) and ({'kd_grad', 'ks_grad'} <= want) and grad_on and not any(is_live(k) for k in smooth)
    if want & set(tex_users):
        with on(*tex_users):
            all_tex = kd_ks.sample(gb_pos_original, idx, mask=tex_mask)
        kd = _I.first_channels(all_tex, 3)           # (the channels the shaded colour carries: a one-pass gradient instead of the slice node's fill + copy)
        ks = all_tex[..., 3:6]
    # Every buffer of the layer is [values, alpha = 1] in the reference (torch.cat((..., alpha), dim=-1) throughout render.py:99-199);
    # the alpha channel is appended by the composite pass, so only the value channels are collected here.
    if fused:
        all_tex_jitter = kd_ks.sample(gb_pos_original + pos_noise, idx, mask=tex_mask)
        nrm_in = (None, None, None, None)
        if 'normal_grad' in want:
            jitter = (util.pixel_grid(W, H, device=dev)[None, ...] + offset).contiguous()
            mask_tap = dr.texture(mask.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
            nrm_jitter = dr.texture(gb_normal.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
            nrm_in = (gb_normal, nrm_jitter, mask, mask_tap)
        kd, out['kd_grad'], out['ks_grad'], ng = _I.material_grads(all_tex, all_tex_jitter, *nrm_in)
        if ng is not None:
            out['normal_grad'] = ng
    elif fused_dead:
        with torch.no_grad():
            all_tex_jitter = kd_ks.sample(gb_pos_original + pos_noise, idx, mask=tex_mask)
            nrm_in = (None, None, None, None)
            if 'normal_grad' in want:
                jitter = (util.pixel_grid(W, H, device=dev)[None, ...] + offset).contiguous()
                mask_tap = dr.texture(mask.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
                nrm_jitter = dr.texture(gb_normal.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
                nrm_in = (gb_normal, nrm_jitter, mask, mask_tap)
            _, out['kd_grad'], out['ks_grad'], ng = _I.material_grads(all_tex, all_tex_jitter, *nrm_in, want_kd=False)
            if ng is not None:
                out['normal_grad'] = ng
    elif want & {'kd_grad', 'ks_grad'}:
        with on('kd_grad', 'ks_grad'):
            all_tex_jitter = kd_ks.sample(gb_pos_original + pos_noise, idx, mask=tex_mask)
            out['kd_grad'] = torch.abs(all_tex_jitter[..., 0:3] - kd)
            ks_w = _const((0.0, 1.0, 1.0), dev)
            out['ks_grad'] = torch.abs(all_tex_jitter[..., 3:6] - ks) * ks_w
    if 'normal_grad' in want and not fused and not fused_dead:
        with on('normal_grad'):
            jitter = (util.pixel_grid(W, H, device=dev)[None, ...] + offset).contiguous()
            mask_tap = dr.texture(mask.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
            nrm_jitter = dr.texture(gb_normal.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
            out['normal_grad'] = torch.abs(nrm_jitter - gb_normal) * (mask * mask_tap)
    if 'normal' in want:
        with on('normal'):
            out['normal'] = ru.prepare_shading_normal(gb_pos, view_pos, None, gb_normal, gb_tangent, gb_geometric_normal, two_sided_shading=True,
                                                      opengl=True)
    if 'shaded' in want:
        out['shaded'] = kd                                         # bsdf = 'kd' (render.py:120,169-170)
        if not fused:
            out['_shaded_of'] = all_tex                            # kd IS all_tex[..., :3]: the fused composite takes the wide tensor (compose)
    if 'kd' in want:
        out['kd'] = kd
    if 'ks' in want:
        out['ks'] = ks
    z_aux, depth_aux, inv_aux = aux
    if 'z_grad' in want:
        out['z_grad'] = z_aux                                      # (z, |dz|, 0): render.py:291-299,105
    if 'geometric_normal' in want:
        out['geometric_normal'] = gb_geometric_normal
    if 'depth' in want:
        if depth_aux is not None:
            out['depth'] = depth_aux
        else:
            out['depth'] = (gb_pos - view_pos).pow(2).sum(dim=-1, keepdim=True).sqrt()
    if 'invdepth' in want:
        if inv_aux is not None:
            out['invdepth'] = inv_aux
        else:
            out['invdepth'] = 1.0 / ((gb_pos - view_pos).pow(2) + 1e-8).sum(dim=-1, keepdim=True).sqrt()
    return out


def render_mesh(FLAGS, idx, ctx, mesh, mesh_original, mtx_in, view_pos, lgt, resolution, spp=1, num_layers=1, msaa=False, background=None,
                optix_ctx=None, bsdf=None, denoiser=None, shadow_scale=1.0, use_uv=True, finetune_normal=True, extra_dict=None, xfm_lgt=None,
                shade_data=False, buffers=None, _keep_rast=False, _rng_draws=None, _grad_buffers=None):
    """`_rng_draws` (extension, tests): {'noise' [B,H,W,3], 'offset' [B,H,W,2], 'pos_noise' [B,H,W,3]} -- the three random tensors of a
    call, which the reference draws from the global generator in this order (render.py:285, :68, :84); None = draw them here.
    `_grad_buffers` (extension, tick_*): the buffers whose gradient somebody will ask for; the others are produced under
    torch.no_grad() and composited / antialiased in a pass of their own, so the backward of a tick that renders all 12 buffers costs
    what the backward of the three to seven it reads costs.  None = every buffer is differentiable, as in the reference."""
    assert num_layers == 1
    spp = int(spp)
    H, W = int(resolution[0]), int(resolution[1])
    Hf, Wf = H * spp, W * spp                     # visibility resolution (render.py:239); == (H, W) in every configuration of train.py (spp = 1)
    want = set(ALL_BUFFERS) if buffers is None else set(buffers)
    want.discard('msdf_image')
    _keep_rast = _keep_rast or (buffers is not None and '_rast' in buffers)
    for k in ('_rast', 'visible_triangles', '_seen_faces'):
        want.discard(k)
    if extra_dict is not None and extra_dict.get('msdf') is not None and (buffers is None or 'msdf_image' in buffers):
        want.add('msdf_image')
    grad_on = torch.is_grad_enabled()
    live = None if (_grad_buffers is None or not grad_on) else (want & set(_grad_buffers))
    view_pos = view_pos[:, None, None, :] if view_pos.dim() == 2 else view_pos
    tri = mesh.t_pos_idx32
    dev = mesh.v_pos.device

    v_pos = _batched(mesh.v_pos)
    v_pos_clip = ru.xfm_points(v_pos, mtx_in)                                      # render.py:396
    B = v_pos_clip.shape[0]
    # the pixel derivatives of the barycentrics are read by the z-gradient pass only (render.py:291-299): not written when no buffer wants it
    no_grad_of = lambda k: k in want and (not grad_on or (live is not None and k not in live))
    need_aux = 'z_grad' in want or no_grad_of('depth') or no_grad_of('invdepth')
    with dr.DepthPeeler(ctx, v_pos_clip, tri, [Hf, Wf]) as peeler:
        rast_full, db_full = peeler.rasterize_next_layer(want_db='z_grad' in want)
    rast, db = rast_full, db_full
    if spp > 1 and msaa:                          # shade at the framebuffer resolution (render.py:241-245): nearest sample of the raster
        rast = util.scale_img_nhwc(rast_full, [H, W], mag='nearest', min='nearest')
        db = util.scale_img_nhwc(db_full, [H, W], mag='nearest', min='nearest') * spp if db_full is not None else None
    elif spp > 1:
        H, W = Hf, Wf                             # no msaa: everything at the visibility resolution, averaged at the end

    F = tri.shape[0]
    # ---- G-buffer: one interpolation pass for everything indexed by t_pos_idx (render.py:257-259,283,328) ------------------
    # one pass over the raster (d3h.raster.gbuffer): each attribute lands in its own contiguous image, attributes no requested buffer
    # reads are neither packed nor produced (nor, for lazily built normals, computed), the face normal (render.py:261-267: an
    # (f, f, f)-indexed interpolation) is a gather by triangle id and the coverage mask of shade() (render.py:66) comes out of the
    # same read
    has_msdf = 'msdf_image' in want
    need_pos, need_nrm = bool(want & {'normal', 'depth', 'invdepth'}), bool(want & {'normal', 'normal_grad'})
    srcs = []
    if need_pos:
        srcs.append(('pos', v_pos))
    srcs.append(('orig', _batched(mesh_original.v_pos)))
    if need_nrm:
        srcs.append(('nrm', _batched(mesh.v_nrm)))
    if has_msdf:
        m = extra_dict['msdf']
        assert m.dim() == 1 or (m.dim() == 2 and m.size(1) == 1)
        srcs.append(('msdf', m.reshape(1, -1, 1)))
    nb_attr = max(t.shape[0] for _, t in srcs)
    packed = torch.cat([t.expand(nb_attr, -1, -1) for _, t in srcs], dim=-1) if len(srcs) > 1 else srcs[0][1]
    fn = _I.face_normals(v_pos, tri) if want & {'geometric_normal', 'normal'} else None      # [B,F,3], one launch
    # (the raster's only differentiable consumer at the shading resolution == visibility resolution is this pass -- antialias and composite take
    # it as a constant, the z / depth pass runs under no_grad --: its backward is folded into the G-buffer's, d3h.raster.gbuffer)
    fold = (H, W) == (Hf, Wf) and os.environ.get('D3H_FUSED_GBUFFER_RASTER_BWD', '1') != '0'
    groups, gb_geometric_normal, cover = _R.gbuffer(packed, [t.shape[-1] for _, t in srcs], rast, tri, face_attr=fn, want_mask=True,
                                                    raster_pos=v_pos_clip if fold else None)
    gb = {k: g for (k, _), g in zip(srcs, groups)}
    gb_pos, gb_pos_original, gb_normal, gb_msdf = gb.get('pos'), gb['orig'], gb.get('nrm'), gb.get('msdf')

    gb_tangent = None
    if 'normal' in want:
        with torch.no_grad():                                                    # render.py:284-287 (use_uv == False branch)
            noise = torch.randn_like(gb_normal) if _rng_draws is None else _rng_draws['noise'].to(dev)
            noise = noise / noise.norm(dim=-1, keepdim=True)
        gb_tangent = torch.cross(noise, gb_normal, dim=-1)

    # forward-only buffers in one fused pass: z / z-gradient (torch.no_grad in the reference, render.py:291-299) and, when nobody
    # differentiates them (the reference does only under FLAGS.use_depth), depth / inverse depth (render.py:197-199)
    aux = (None, None, None)
    if need_aux:
        aux = _R.aux_buffers(v_pos_clip, rast, db, tri, gb_pos, view_pos, want_z='z_grad' in want, want_depth=no_grad_of('depth'),
                             want_invdepth=no_grad_of('invdepth'))

    layer = shade(FLAGS, idx, rast, aux, gb_pos, gb_pos_original, gb_geometric_normal, gb_normal, gb_tangent, view_pos, mesh.material,
                  want, finetune_normal, mask=cover, rng_draws=_rng_draws, live=live, skip_uncovered=(H, W) == (Hf, Wf))
    shaded_of = layer.pop('_shaded_of', None)
    if (H, W) != (Hf, Wf):
        shaded_of = None
    if has_msdf:
        layer['msdf_image'] = gb_msdf
    if (H, W) != (Hf, Wf):                        # back up to the visibility resolution (render.py:334-336)
        layer = {k: (util.scale_img_nhwc(t, [Hf, Wf], mag='nearest', min='nearest') if t is not None else None) for k, t in layer.items()}

    # ---- composite against each buffer's background (one pass), then ONE antialias pass over all channels (render.py:375-382,430-449)
    if background is None:
        background = torch.zeros(1, Hf, Wf, 3, dtype=torch.float32, device=dev)
    elif spp > 1:
        background = util.scale_img_nhwc(background, [Hf, Wf], mag='nearest', min='nearest')          # render.py:424-425

    def compose(keys):
        sources = []
        for k in keys:
            if k == 'shaded':
                sources.append((layer[k], _I.COMP_IMAGE, background))
            elif k == 'depth':
                sources.append((layer[k], _I.COMP_CONST20, None))
            elif k == 'msdf_image':                     # lerp(0, 1, coverage * msdf): the value IS the alpha (render.py:444-449)
                sources.append((layer[k], _I.COMP_ALPHA, None))
            else:
                sources.append((layer[k], _I.COMP_ZERO, None))
        widths = [1 if k == 'msdf_image' else layer[k].shape[-1] + 1 for k in keys]
        if not torch.is_grad_enabled() and os.environ.get('D3H_FUSED_COMPOSITE_AA', '1') != '0':
            # nobody differentiates this pass (dead buffers, validation renders): composite + antialias as ONE forward kernel
            img = _I.composite_antialias(rast_full, sources, v_pos_clip, tri)
        elif os.environ.get('D3H_FUSED_COMPOSITE_AA_GRAD', '1') != '0':
            # the differentiated pass: one kernel each way, the composited pre-antialias image is never stored (round 6); the shaded colour
            # goes in as the texture MLP's whole output + "first three channels", so that its gradient comes back at that width in the same pass
            if shaded_of is not None and os.environ.get('D3H_FUSED_COMPOSITE_PREFIX', '1') != '0':
                sources = [(shaded_of, kind, bg_, 3) if k == 'shaded' else (s_, kind, bg_) for k, (s_, kind, bg_) in zip(keys, sources)]
            img = _I.composite_antialias_grad(rast_full, sources, v_pos_clip, tri)
        else:
            img = dr.antialias(_I.composite(rast_full, sources), rast_full, v_pos_clip, tri)
        return (util.avg_pool_nhwc(img, spp) if spp > 1 else img), widths                              # render.py:449

    all_keys = [k for k in list(ALL_BUFFERS) + ['msdf_image'] if k in layer]
    live_keys = all_keys if live is None else [k for k in all_keys if k in live]
    dead_keys = [k for k in all_keys if k not in live_keys]
    # '_stacked' / '_layout': the channel-concatenated image itself, for consumers that read several buffers in one pass
    # (d3h.imgops.pixel_losses); the per-buffer entries are views of it, as the reference's separate tensors would be
    out_buffers = {'_layout': {}}
    for name, keys, no_grad in (('_stacked', live_keys, False), ('_stacked_nograd', dead_keys, True)):
        if not keys:
            continue
        with torch.set_grad_enabled(grad_on and not no_grad):
            stacked, widths = compose(keys)
        out_buffers[name] = stacked
        c0 = 0
        for k, n in zip(keys, widths):
            out_buffers[k] = stacked[..., c0:c0 + n]
            if not no_grad:
                out_buffers['_layout'][k] = (c0, n)
            c0 += n
    if '_stacked' not in out_buffers:
        out_buffers['_stacked'] = None
    if _keep_rast:
        out_buffers['_rast'] = rast_full
    if buffers is None or 'visible_triangles' in buffers or '_seen_faces' in buffers:
        # render.py:404-407 -- sorted unique triangle ids; bitmap scatter + nonzero instead of sorting a million ids.  '_seen_faces'
        # is the bitmap itself: consumers that only need "is this triangle visible" avoid nonzero's host synchronisation
        seen = torch.zeros(F + 1, dtype=torch.bool, device=dev)
        # index_fill_, not `seen[ids] = True`: the indexed assignment uploads the Python scalar as a tensor, which synchronises the stream
        seen.index_fill_(0, rast_full[..., 3].reshape(-1).long(), True)
        out_buffers['_seen_faces'] = seen[1:]
        if buffers is None or 'visible_triangles' in buffers:
            out_buffers['visible_triangles'] = torch.nonzero(seen[1:]).reshape(-1)          # a plain int64 tensor, as the reference's
    return out_buffers


def render_uv(ctx, mesh, resolution, mlp_texture):
    """Texture bake of the export step (render.py:456-472; called by train.py:218 after xatlas): the mesh rasterised in its uv chart,
    world positions interpolated per texel, kd / ks sampled from the texture MLP.  -> (coverage [1,H,W,1], kd [1,H,W,3], ks [1,H,W,3])"""
    uv = mesh.v_tex[None, ...] * 2.0 - 1.0
    clip = torch.cat((uv, torch.zeros_like(uv[..., 0:1]), torch.ones_like(uv[..., 0:1])), dim=-1).contiguous()
    rast, _ = dr.rasterize(ctx, clip, mesh.t_tex_idx.int(), resolution)
    v_pos = mesh.v_pos if mesh.v_pos.dim() == 3 else mesh.v_pos[None, ...]
    gb_pos, _ = interpolate(v_pos, rast, mesh.t_pos_idx.int())
    cover = (rast[..., -1:] > 0).float()
    tex = mlp_texture.sample(gb_pos)                    # every texel, as the reference (uncovered texels are dilated over by the caller)
    assert tex.shape[-1] == 6, "Combined kd_ks must be 6 channels"
    return cover, tex[..., 0:3], tex[..., 3:6]
