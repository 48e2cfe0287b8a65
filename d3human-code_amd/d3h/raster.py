"""Host side of csrc/raster.hip: rasterize / interpolate / antialias / texture with autograd.

Signatures follow nvdiffrast.torch as the reference calls it (render/render.py:37,72,102,381,400-403); the shim module
`nvdiffrast/torch.py` re-exports these.  Tensors: pos [B or 1, V, 4] clip space, tri [F,3] int32, images NHWC float32.
"""
import os
import torch

from . import _lib as L


# The tile-binned rasteriser (csrc/raster.hip: raster_tile_kernel; north_star: "tile-binned differentiable rasterizer") is BUILT, bit-identical to
# the wave-per-triangle kernels and NOT selected by default: measured on four 1024^2 frames (profiles/r5_raster_vs_triangles.txt) it is slower at
# every mesh size tried -- 9 k / 37 k / 148 k / 593 k triangles: 181 / 313 / 677 / 2 333 us against 70 / 87 / 181 / 524 us.  Three passes over the
# triangles (count, fill, per-tile) with gathered vertex loads cost more than the one wave-uniform pass they replace, and the covered 11 % of the
# tiles carry all the work.  D3H_RASTER_BIN_MIN=<triangles> selects it from that mesh size on (tests force it with BIN_MIN_TRIS = 1).
BIN_MIN_TRIS = int(os.environ.get('D3H_RASTER_BIN_MIN', str(1 << 30)))
BIN_PAIRS_PER_TRI = 4


def _bstride(t):
    """batch stride in elements, 0 for a broadcast batch of 1"""
    return 0 if t.shape[0] == 1 else t.shape[1] * t.shape[2]


class _Scratch:
    """per-device reusable scratch buffers (z-buffer, big-triangle list, edge hash)"""
    bufs = {}

    @classmethod
    def get(cls, name, nbytes, dev):
        key = (name, str(dev))
        b = cls.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            cls.bufs[key] = b
        return b


class _RasterizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W, nb, want_db=True):
        lib = L.lib()
        pos_c = pos.contiguous().float()
        dev = pos.device
        nv, nf = pos_c.shape[1], tri.shape[0]
        rast = torch.empty(nb, H, W, 4, dtype=torch.float32, device=dev)
        # (want_db False: the caller reads no pixel derivatives -- 16 bytes per pixel the resolve pass does not write; an empty tensor comes back)
        db = torch.empty((nb, H, W, 4) if want_db else (0,), dtype=torch.float32, device=dev)
        zbuf = _Scratch.get('zbuf', nb * H * W * 8, dev)
        big, big_cap = None, 0
        if nf >= BIN_MIN_TRIS:
            # tile-binned rasteriser (csrc/raster.hip: raster_tile_kernel): header + room for BIN_PAIRS_PER_TRI (triangle, tile) pairs per triangle
            # and frame; a render that needs more falls back on the device to the wave-per-triangle kernels -- no host read-back
            nt = nb * (-(-W // 32)) * (-(-H // 32))
            big_cap = min(3 * nt + 8 + BIN_PAIRS_PER_TRI * nf * nb + nf, (1 << 31) - 1)
            big = _Scratch.get('bins', 4 * big_cap, dev).view(torch.int32)
        L.check(lib.d3h_rasterize_fwd(L.ptr(pos_c), L.i32(nv), L.i32(_bstride(pos_c)), L.ptr(tri), L.i32(nf), L.i32(nb), L.i32(H), L.i32(W),
                                      L.ptr(zbuf), L.ptr(big), L.i32(big_cap), L.ptr(rast), L.ptr(db if want_db else None), L.stream()), 'rasterize_fwd')
        ctx.save_for_backward(pos_c, tri, rast)
        ctx.dims = (H, W, nb)
        ctx.mark_non_differentiable(db)
        # (without this the engine hands the backward a zero-filled [nb, H, W, 4] tensor for `db` on every step: a 67 MB fill nobody reads)
        ctx.set_materialize_grads(False)
        ctx.zeros = L.zeros_like(pos) if ctx.needs_input_grad[0] else None      # d_pos, filled ahead of the backward (d3h/mtets.py)
        return rast, db

    @staticmethod
    def backward(ctx, g_rast, _g_db):
        pos, tri, rast = ctx.saved_tensors
        H, W, nb = ctx.dims
        d_pos, ctx.zeros = getattr(ctx, 'zeros', None), None
        if d_pos is None:
            d_pos = L.zeros_like(pos)
        if g_rast is None:                     # nothing flowed into the barycentrics: the position gradient through them is zero
            return d_pos, None, None, None, None, None
        L.check(L.lib().d3h_rasterize_bwd(L.ptr(pos), L.i32(_bstride(pos)), L.ptr(tri), L.i32(nb), L.i32(H), L.i32(W), L.ptr(rast),
                                          L.ptr(g_rast.contiguous()), L.ptr(d_pos), L.stream()), 'rasterize_bwd')
        return d_pos, None, None, None, None, None


def rasterize(pos, tri, resolution, nb=None, want_db=True):
    """-> (rast [B,H,W,4] = (u, v, z/w, tri_id+1), rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY); want_db False (extension): rast_db is None)"""
    H, W = int(resolution[0]), int(resolution[1])
    nb = pos.shape[0] if nb is None else nb
    rast, db = _RasterizeFn.apply(pos, tri.contiguous(), H, W, nb, bool(want_db))
    return rast, (db if want_db else None)


class _InterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db):
        lib = L.lib()
        attr_c = attr.contiguous().float()
        rast_c = rast.contiguous()
        nb, H, W = rast_c.shape[:3]
        na = attr_c.shape[2]
        out = torch.empty(nb, H, W, na, dtype=torch.float32, device=attr.device)
        out_da = torch.empty(nb, H, W, 2 * na, dtype=torch.float32, device=attr.device) if rast_db is not None else None
        L.check(lib.d3h_interpolate_fwd(L.ptr(attr_c), L.i32(_bstride(attr_c)), L.i32(na), L.ptr(rast_c), L.ptr(tri),
                                        L.ptr(rast_db.contiguous() if rast_db is not None else None), L.i32(nb), L.i32(H), L.i32(W),
                                        L.ptr(out), L.ptr(out_da), L.stream()), 'interpolate_fwd')
        ctx.save_for_backward(attr_c, rast_c, tri)
        if out_da is None:
            out_da = out.new_empty(0)
        ctx.mark_non_differentiable(out_da)
        return out, out_da

    @staticmethod
    def backward(ctx, g_out, _g_da):
        attr, rast, tri = ctx.saved_tensors
        nb, H, W = rast.shape[:3]
        na = attr.shape[2]
        need_attr, need_rast = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_attr = L.zeros_like(attr) if need_attr else None
        d_rast = torch.empty_like(rast) if need_rast else None
        L.check(L.lib().d3h_interpolate_bwd(L.ptr(attr), L.i32(_bstride(attr)), L.i32(na), L.ptr(rast), L.ptr(tri), L.ptr(g_out.contiguous()),
                                            L.i32(nb), L.i32(H), L.i32(W), L.ptr(d_attr), L.ptr(d_rast), L.stream()), 'interpolate_bwd')
        return d_attr, d_rast, None, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """nvdiffrast.interpolate: (out [B,H,W,A], out_da [B,H,W,2A] or empty).  The attribute pixel derivatives are
    forward-only: the reference only evaluates them under no_grad (render/render.py:291-299)."""
    if attr.dim() == 2:
        attr = attr[None]
    out, da = _InterpolateFn.apply(attr, rast, tri.contiguous(), rast_db if diff_attrs is not None else None)
    return out, (da if da.numel() else None)


class _GBufferFn(torch.autograd.Function):
    """every interpolation of one render layer in one pass (csrc/raster.hip: gbuffer_*): the vertex-attribute groups of a packed
    attribute array, a per-face attribute gathered by triangle id, and the coverage mask"""

    @staticmethod
    def forward(ctx, attr, face_attr, rast, tri, widths, need, want_mask, pos):
        lib = L.lib()
        attr_c = attr.contiguous().float()
        rast_c = rast.contiguous()
        nb, H, W = rast_c.shape[:3]
        na = attr_c.shape[2]
        dev = attr.device
        w4 = list(widths) + [0] * (4 - len(widths))
        outs = [torch.empty(nb, H, W, w, dtype=torch.float32, device=dev) if (k < len(widths) and need[k]) else None for k, w in enumerate(w4)]
        fa = face_attr.contiguous().float() if face_attr is not None else None
        fw = fa.shape[2] if fa is not None else 0
        face_out = torch.empty(nb, H, W, fw, dtype=torch.float32, device=dev) if fa is not None else None
        if fa is not None and fa.shape[1] == 0:            # a mesh without faces covers nothing: zeros, and no pointer to gather from
            face_out.zero_()
        mask = torch.empty(nb, H, W, 1, dtype=torch.float32, device=dev) if want_mask else None
        L.check(lib.d3h_gbuffer_fwd(L.ptr(attr_c), L.i32(_bstride(attr_c)), L.i32(na), L.ptr(fa), L.i32(_bstride(fa) if fa is not None else 0),
                                    L.i32(fw), L.ptr(rast_c), L.ptr(tri), L.i32(nb), L.i32(H), L.i32(W), L.ptr(outs[0]), L.i32(w4[0]),
                                    L.ptr(outs[1]), L.i32(w4[1]), L.ptr(outs[2]), L.i32(w4[2]), L.ptr(outs[3]), L.i32(w4[3]),
                                    L.ptr(face_out if (fa is not None and fa.shape[1] > 0) else None), L.ptr(mask), L.stream()), 'gbuffer_fwd')
        # `pos` (the clip positions `rast` was rasterised from; rast itself then arrives detached): the rasteriser's backward runs inside this
        # node's backward pass -- d(barycentrics) never leaves the kernel (csrc/raster.hip: gbuffer_bwd_kernel<true>)
        pos_c = pos.contiguous().float() if pos is not None else None
        ctx.save_for_backward(attr_c, rast_c, tri, pos_c)
        ctx.meta = (w4, fa.shape if fa is not None else None)
        ctx.set_materialize_grads(False)       # outputs nobody differentiates arrive as None (the backward skips them), not as zero-filled images
        empty = attr_c.new_empty(0)
        ret = [o if o is not None else empty for o in outs[:len(widths)]]
        ret.append(face_out if face_out is not None else empty)
        ret.append(mask if mask is not None else empty)
        ctx.mark_non_differentiable(ret[-1])
        return tuple(ret)

    @staticmethod
    def backward(ctx, *gs):
        attr, rast, tri, pos = ctx.saved_tensors
        w4, fshape = ctx.meta
        nb, H, W = rast.shape[:3]
        na = attr.shape[2]
        ng = len(gs) - 2
        g4 = [None] * 4
        for k in range(ng):
            if gs[k] is not None and gs[k].numel():
                g4[k] = gs[k].contiguous().float()
        g_face = gs[ng].contiguous().float() if (gs[ng] is not None and gs[ng].numel() and fshape is not None and fshape[1] > 0) else None
        d_attr = L.zeros_like(attr) if ctx.needs_input_grad[0] else None
        d_face = L.zeros(fshape, torch.float32, attr.device) if (fshape is not None and ctx.needs_input_grad[1] and g_face is not None) else None
        fb = (fshape[1] * fshape[2] if fshape[0] > 1 else 0) if fshape is not None else 0
        if pos is not None and ctx.needs_input_grad[7]:
            d_pos = L.zeros_like(pos)
            L.check(L.lib().d3h_gbuffer_raster_bwd(L.ptr(attr), L.i32(_bstride(attr)), L.i32(na), L.i32(fb), L.i32(fshape[2] if fshape is not None else 0),
                                                   L.ptr(rast), L.ptr(tri), L.i32(nb), L.i32(H), L.i32(W), L.ptr(g4[0]), L.i32(w4[0]), L.ptr(g4[1]),
                                                   L.i32(w4[1]), L.ptr(g4[2]), L.i32(w4[2]), L.ptr(g4[3]), L.i32(w4[3]), L.ptr(g_face), L.ptr(d_attr),
                                                   L.ptr(d_face), L.ptr(pos), L.i32(_bstride(pos)), L.ptr(d_pos), L.stream()), 'gbuffer_raster_bwd')
            return d_attr, d_face, None, None, None, None, None, d_pos
        d_rast = torch.empty_like(rast) if ctx.needs_input_grad[2] else None
        L.check(L.lib().d3h_gbuffer_bwd(L.ptr(attr), L.i32(_bstride(attr)), L.i32(na), L.i32(fb), L.i32(fshape[2] if fshape is not None else 0),
                                        L.ptr(rast), L.ptr(tri), L.i32(nb), L.i32(H), L.i32(W), L.ptr(g4[0]), L.i32(w4[0]), L.ptr(g4[1]),
                                        L.i32(w4[1]), L.ptr(g4[2]), L.i32(w4[2]), L.ptr(g4[3]), L.i32(w4[3]), L.ptr(g_face), L.ptr(d_attr),
                                        L.ptr(d_face), L.ptr(d_rast), L.stream()), 'gbuffer_bwd')
        return d_attr, d_face, d_rast, None, None, None, None, None


def gbuffer(attr, widths, rast, tri, need=None, face_attr=None, want_mask=True, raster_pos=None):
    """attr [B or 1, V, sum(widths)] -> (list of [B,H,W,w_k] (None where need[k] is False), face image [B,H,W,fw] or None,
    mask [B,H,W,1] or None).  Replaces one dr.interpolate per attribute + the (f, f, f)-indexed face-normal interpolation +
    `rast[..., -1:] > 0` of render/render.py:257-267,283,328,66."""
    if attr.dim() == 2:
        attr = attr[None]
    if face_attr is not None and face_attr.dim() == 2:
        face_attr = face_attr[None]
    widths = tuple(int(w) for w in widths)
    assert 1 <= len(widths) <= 4 and sum(widths) == attr.shape[-1]
    need = tuple(bool(n) for n in (need if need is not None else [True] * len(widths)))
    # raster_pos: the clip positions `rast` = rasterize(raster_pos, tri) came from, for a raster with NO other differentiable consumer: the
    # rasteriser's backward then runs inside this op's backward pass (d(barycentrics) is never written out) and `rast` is cut from the graph
    if raster_pos is not None and torch.is_grad_enabled() and raster_pos.requires_grad and rast.requires_grad:
        r = _GBufferFn.apply(attr, face_attr, rast.detach(), tri.contiguous(), widths, need, bool(want_mask), raster_pos)
    else:
        r = _GBufferFn.apply(attr, face_attr, rast, tri.contiguous(), widths, need, bool(want_mask), None)
    groups = [r[k] if need[k] else None for k in range(len(widths))]
    return groups, (r[len(widths)] if face_attr is not None else None), (r[len(widths) + 1] if want_mask else None)


def aux_buffers(clip, rast, db, tri, gb_pos, view_pos, want_z=True, want_depth=True, want_invdepth=True):
    """forward-only z_grad [B,H,W,3] (render.py:291-299), depth / invdepth [B,H,W,1] (render.py:197-199) in one pass -- the buffers of a
    render layer that carry no gradient.  clip [B,V,4]; rast, db [B,H,W,4]; gb_pos [B,H,W,3]; view_pos [B or 1, 1, 1, 3] (or [B,3])"""
    nb, H, W = rast.shape[:3]
    dev = rast.device
    with torch.no_grad():
        clip_c = clip.detach().contiguous().float() if want_z else None
        z = torch.empty(nb, H, W, 3, dtype=torch.float32, device=dev) if want_z else None
        dep = torch.empty(nb, H, W, 1, dtype=torch.float32, device=dev) if want_depth else None
        inv = torch.empty(nb, H, W, 1, dtype=torch.float32, device=dev) if want_invdepth else None
        gp = vp = None
        vstride = 0
        if want_depth or want_invdepth:
            gp = gb_pos.detach().contiguous().float()
            vp = view_pos.detach().reshape(-1, 3).contiguous().float()
            vstride = 3 if vp.shape[0] > 1 else 0
            assert vp.shape[0] in (1, nb)
        L.check(L.lib().d3h_aux_buffers_fwd(L.ptr(clip_c), L.i32(_bstride(clip_c) if want_z else 0), L.ptr(rast.contiguous()),
                                            L.ptr(db.contiguous()) if want_z else None, L.ptr(tri.contiguous()) if want_z else None, L.ptr(gp), L.ptr(vp),
                                            L.i32(vstride), L.i32(nb), L.i32(H), L.i32(W), L.ptr(z), L.ptr(dep), L.ptr(inv), L.stream()), 'aux_buffers_fwd')
    return z, dep, inv


def _hash_for(tri):
    nf = tri.shape[0]
    cap = 1024
    while cap < 12 * max(nf, 1):
        cap *= 2
    dev = tri.device
    kv = _Scratch.get('aa_keys_vals', cap * 16, dev)          # keys [cap] uint64 | vals [2 cap] int32, contiguous: the library fills both at once
    keys, vals = kv[:cap * 8], kv[cap * 8:cap * 16]
    L.check(L.lib().d3h_antialias_hash(L.ptr(tri), L.i32(nf), L.ptr(keys), L.ptr(vals), L.i32(cap), L.stream()), 'antialias_hash')
    return keys, vals, cap


def _edge_flags(pos_c, tri, nb, H, W):
    """[nb, nf] uint8: the silhouette-edge bits of every triangle in every frame (csrc/raster.hip:aa_edge_flags_kernel)"""
    nf = tri.shape[0]
    keys, vals, cap = _hash_for(tri)
    flags = torch.empty(nb, max(nf, 1), dtype=torch.uint8, device=pos_c.device)
    L.check(L.lib().d3h_antialias_flags(L.ptr(pos_c), L.i32(_bstride(pos_c)), L.ptr(tri), L.i32(nf), L.i32(nb), L.ptr(keys), L.ptr(vals), L.i32(cap),
                                        L.i32(H), L.i32(W), L.ptr(flags), L.stream()), 'antialias_flags')
    return flags


class _AntialiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri):
        lib = L.lib()
        color_c, rast_c, pos_c = color.contiguous().float(), rast.contiguous(), pos.contiguous().float()
        nb, H, W, C = color_c.shape
        flags = _edge_flags(pos_c, tri, nb, H, W)        # per render; the backward reuses them (nb x nf bytes)
        out = torch.empty_like(color_c)
        L.check(lib.d3h_antialias_fwd(L.ptr(color_c), L.ptr(rast_c), L.ptr(pos_c), L.i32(_bstride(pos_c)), L.ptr(tri), L.i32(tri.shape[0]),
                                      L.ptr(flags), L.i32(nb), L.i32(H), L.i32(W), L.i32(C), L.ptr(out), L.stream()), 'antialias_fwd')
        ctx.save_for_backward(color_c, rast_c, pos_c, tri, flags)
        return out

    @staticmethod
    def backward(ctx, g_out):
        color, rast, pos, tri, flags = ctx.saved_tensors
        nb, H, W, C = color.shape
        g_color = torch.empty_like(color)
        d_pos = L.zeros_like(pos) if ctx.needs_input_grad[2] else None
        L.check(L.lib().d3h_antialias_bwd(L.ptr(color), L.ptr(rast), L.ptr(pos), L.i32(_bstride(pos)), L.ptr(tri), L.i32(tri.shape[0]), L.ptr(flags),
                                          L.i32(nb), L.i32(H), L.i32(W), L.i32(C), L.ptr(g_out.contiguous()), L.ptr(g_color), L.ptr(d_pos),
                                          L.stream()), 'antialias_bwd')
        return g_color, None, d_pos, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    return _AntialiasFn.apply(color, rast, pos, tri.contiguous())


class _TextureFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv):
        tex_c, uv_c = tex.contiguous().float(), uv.contiguous().float()
        nb, H, W = uv_c.shape[:3]
        TH, TW, C = tex_c.shape[1:]
        out = torch.empty(nb, H, W, C, dtype=torch.float32, device=tex.device)
        bs = 0 if tex_c.shape[0] == 1 else TH * TW * C
        L.check(L.lib().d3h_texture_fwd(L.ptr(tex_c), L.i32(bs), L.i32(TH), L.i32(TW), L.i32(C), L.ptr(uv_c), L.i32(nb), L.i32(H), L.i32(W),
                                        L.ptr(out), L.stream()), 'texture_fwd')
        ctx.save_for_backward(uv_c)
        ctx.tshape = tuple(tex_c.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (uv,) = ctx.saved_tensors
        nb, H, W = uv.shape[:3]
        B, TH, TW, C = ctx.tshape
        d_tex = L.zeros(ctx.tshape, torch.float32, uv.device)
        bs = 0 if B == 1 else TH * TW * C
        L.check(L.lib().d3h_texture_bwd(L.i32(bs), L.i32(TH), L.i32(TW), L.i32(C), L.ptr(uv), L.i32(nb), L.i32(H), L.i32(W),
                                        L.ptr(g_out.contiguous()), L.ptr(d_tex), L.stream()), 'texture_bwd')
        return d_tex, None


def texture(tex, uv, filter_mode='linear', boundary_mode='clamp', **kw):
    """nvdiffrast.texture for the one mode the reference uses (render/render.py:72,102): bilinear, clamp; no uv gradient
    (the jittered lookup coordinates are constants)."""
    if filter_mode != 'linear' or boundary_mode != 'clamp':
        raise NotImplementedError('d3h.texture: only filter_mode="linear", boundary_mode="clamp"')
    return _TextureFn.apply(tex, uv)
