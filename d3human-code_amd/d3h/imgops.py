"""Host side of csrc/image_ops.hip: mesh normals, shading normal, image loss, SSIM, SDF edge regulariser (autograd)."""
import ctypes
import os

import torch
from .devconst import const as _const

from . import _lib as L


# ---- mesh normals --------------------------------------------------------------------------------------
class _AutoNormalsFn(torch.autograd.Function):
    """v: [P,3] or [B,P,3] (B vertex sets sharing the face list)"""

    @staticmethod
    def forward(ctx, v, f32):
        v = v.contiguous().float()
        nb = 1 if v.dim() == 2 else v.shape[0]
        nv, nf = v.shape[-2], f32.shape[0]
        raw = torch.empty_like(v)
        vn = torch.empty_like(v)
        L.check(L.lib().d3h_auto_normals_fwd(L.ptr(v), L.i32(nb), L.i32(nv), L.ptr(f32), L.i32(nf), L.ptr(raw), L.ptr(vn), L.stream()),
                'auto_normals_fwd')
        ctx.save_for_backward(v, f32, raw)
        return vn

    @staticmethod
    def backward(ctx, g):
        v, f32, raw = ctx.saved_tensors
        nb = 1 if v.dim() == 2 else v.shape[0]
        d_v = L.zeros_like(v)
        g_raw = torch.empty_like(v)
        L.check(L.lib().d3h_auto_normals_bwd(L.ptr(v), L.i32(nb), L.i32(v.shape[-2]), L.ptr(f32), L.i32(f32.shape[0]), L.ptr(raw),
                                             L.ptr(g.contiguous()), L.ptr(g_raw), L.ptr(d_v), L.stream()), 'auto_normals_bwd')
        return d_v, None


def auto_normals(v_pos, faces32):
    """render/mesh.py:418-446: area-weighted, normalised vertex normals; zero-length -> (0,0,1).  v_pos [P,3] or [B,P,3]."""
    return _AutoNormalsFn.apply(v_pos, faces32.contiguous())


class _FaceNormalsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, f32):
        v = v.contiguous().float()
        nb = 1 if v.dim() == 2 else v.shape[0]
        nf = f32.shape[0]
        fn = torch.empty(*v.shape[:-2], nf, 3, dtype=torch.float32, device=v.device)
        L.check(L.lib().d3h_face_normals_fwd(L.ptr(v), L.i32(nb), L.i32(v.shape[-2]), L.ptr(f32), L.i32(nf), L.ptr(fn), L.stream()),
                'face_normals_fwd')
        ctx.save_for_backward(v, f32)
        ctx.zeros = L.zeros_like(v) if ctx.needs_input_grad[0] else None      # d_v, filled ahead of the backward (d3h/mtets.py)
        return fn

    @staticmethod
    def backward(ctx, g):
        v, f32 = ctx.saved_tensors
        nb = 1 if v.dim() == 2 else v.shape[0]
        d_v, ctx.zeros = getattr(ctx, 'zeros', None), None
        if d_v is None:
            d_v = L.zeros_like(v)
        L.check(L.lib().d3h_face_normals_bwd(L.ptr(v), L.i32(nb), L.i32(v.shape[-2]), L.ptr(f32), L.i32(f32.shape[0]), L.ptr(g.contiguous()),
                                             L.ptr(d_v), L.stream()), 'face_normals_bwd')
        return d_v, None


def face_normals(v_pos, faces32):
    """render/render.py:261-264: safe_normalize(cross(v1 - v0, v2 - v0)) per face.  v_pos [P,3] or [B,P,3]."""
    return _FaceNormalsFn.apply(v_pos, faces32.contiguous())


# ---- prepare_shading_normal ---------------------------------------------------------------------------
def _bc_strides(ts, shape):
    B, H, W = shape
    out = []
    keep = []
    for t in ts:
        t = t.float()
        while t.dim() < 4:
            t = t[None]
        t = t.contiguous()
        keep.append(t)
        st = [t.stride(k) if t.shape[k] != 1 else 0 for k in range(3)]
        for k, full in enumerate((B, H, W)):
            if t.shape[k] not in (1, full):
                raise RuntimeError('prepare_shading_normal: shapes are not broadcastable')
        out += st
    arr = (ctypes.c_int64 * 18)(*out)
    return keep, arr


class _ShadingNormalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, view_pos, pert, snrm, stng, gnrm, two_sided, opengl):
        ins = [pos, view_pos, pert, snrm, stng, gnrm]
        shp = [max(t.shape[k] if t.dim() == 4 else 1 for t in ins) for k in range(3)]
        keep, strides = _bc_strides(ins, shp)
        B, H, W = shp
        out = torch.empty(B, H, W, 3, dtype=torch.float32, device=pos.device)
        L.check(L.lib().d3h_shading_normal_fwd(*[L.ptr(t) for t in keep], strides, L.i32(B), L.i32(H), L.i32(W), L.i32(int(two_sided)),
                                               L.i32(int(opengl)), L.ptr(out), L.stream()), 'shading_normal_fwd')
        ctx.save_for_backward(*keep)
        ctx.meta = (shp, bool(two_sided), bool(opengl), [tuple(t.shape) for t in ins])
        return out

    @staticmethod
    def backward(ctx, g):
        keep = list(ctx.saved_tensors)
        shp, two_sided, opengl, in_shapes = ctx.meta
        B, H, W = shp
        _, strides = _bc_strides(keep, shp)
        grads = [torch.empty(B, H, W, 3, dtype=torch.float32, device=g.device) for _ in range(6)]
        L.check(L.lib().d3h_shading_normal_bwd(*[L.ptr(t) for t in keep], strides, L.i32(B), L.i32(H), L.i32(W), L.i32(int(two_sided)),
                                               L.i32(int(opengl)), L.ptr(g.contiguous()), *[L.ptr(t) for t in grads], L.stream()),
                'shading_normal_bwd')
        outs = []
        for gr, shape in zip(grads, in_shapes):
            full = (1,) * (4 - len(shape)) + tuple(shape)
            red = [k for k in range(3) if full[k] == 1 and shp[k] != 1]
            if red:
                gr = gr.sum(dim=red, keepdim=True)
            outs.append(gr.reshape(shape))
        return (*outs, None, None)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True):
    if perturbed_nrm is None:   # renderutils/ops.py:220-221
        perturbed_nrm = _const((0.0, 0.0, 1.0), pos.device)[None, None, None, :]
    return _ShadingNormalFn.apply(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl)


# ---- image loss -------------------------------------------------------------------------------------------
_LOSS = {'l1': 0, 'mse': 1, 'smape': 2, 'relmse': 3}
_TONE = {'none': 0, 'log_srgb': 1}


class _ImageLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, target, loss, tonemapper):
        img_c = img.float().expand(torch.broadcast_shapes(img.shape, target.shape)).contiguous()
        tgt_c = target.float().expand(img_c.shape).contiguous()
        npix = img_c.numel() // 3
        out = torch.empty(1, dtype=torch.float32, device=img.device)
        L.check(L.lib().d3h_image_loss_fwd(L.ptr(img_c), L.ptr(tgt_c), L.i64(npix), L.i32(_LOSS[loss]), L.i32(_TONE[tonemapper]), L.ptr(out),
                                           L.stream()), 'image_loss_fwd')
        ctx.save_for_backward(img_c, tgt_c)
        ctx.meta = (loss, tonemapper, npix, img.shape, target.shape)
        return out[0] / npix

    @staticmethod
    def backward(ctx, g):
        img_c, tgt_c = ctx.saved_tensors
        loss, tonemapper, npix, ishape, tshape = ctx.meta
        d_img = torch.empty_like(img_c) if ctx.needs_input_grad[0] else None
        d_tgt = torch.empty_like(tgt_c) if ctx.needs_input_grad[1] else None
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_image_loss_bwd(L.ptr(img_c), L.ptr(tgt_c), L.i64(npix), L.i32(_LOSS[loss]), L.i32(_TONE[tonemapper]), L.ptr(gs),
                                           L.f32(1.0 / npix), L.ptr(d_img), L.ptr(d_tgt), L.stream()), 'image_loss_bwd')
        red = lambda t, s: None if t is None else t.sum_to_size(s)
        return red(d_img, ishape), red(d_tgt, tshape), None, None


def image_loss(img, target, loss='l1', tonemapper='none'):
    """renderutils/ops.py:479-501 (the live CUDA path, loss.cu:95): scalar mean over pixels of the channel-mean loss"""
    return _ImageLossFn.apply(img, target, loss, tonemapper)


# ---- SSIM ----------------------------------------------------------------------------------------------------
class _SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a_c, b_c = a.contiguous().float(), b.contiguous().float()
        H, W = a_c.shape[-2:]
        N = a_c.numel() // (H * W)
        need = a.requires_grad or b.requires_grad
        tmp = None
        gmom = torch.empty(5 * N * H * W, dtype=torch.float32, device=a.device) if need else None
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        need_b = bool(b.requires_grad)          # a constant second image (the target of a loss): three moment-gradient planes instead of five
        L.check(L.lib().d3h_ssim_fwd(L.ptr(a_c), L.ptr(b_c), L.i32(N), L.i32(H), L.i32(W), L.ptr(tmp), L.ptr(gmom), L.i32(need_b), L.ptr(out), None, L.i32(1),
                                     L.stream()), 'ssim_fwd')
        if need:
            ctx.save_for_backward(a_c, b_c, gmom)
        ctx.dims = (N, H, W)
        ctx.need_b = need_b
        return out[0] / (N * H * W)

    @staticmethod
    def backward(ctx, g):
        a_c, b_c, gmom = ctx.saved_tensors
        N, H, W = ctx.dims
        tmp = None
        d_a = torch.empty_like(a_c) if ctx.needs_input_grad[0] else None
        d_b = torch.empty_like(b_c) if (ctx.needs_input_grad[1] and ctx.need_b) else None
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_ssim_bwd(L.ptr(a_c), L.ptr(b_c), L.i32(N), L.i32(H), L.i32(W), L.ptr(gmom), L.i32(ctx.need_b), L.ptr(tmp), L.ptr(gs),
                                     L.f32(1.0 / (N * H * W)), L.ptr(d_a), L.ptr(d_b), None, L.i32(1), L.stream()), 'ssim_bwd')
        return d_a, d_b


def ssim(img1, img2, window_size=11, size_average=True):
    """ssim_loss.py:33-63 for [B,C,H,W] (or [C,H,W]) images: mean SSIM with an 11x11 Gaussian window (sigma 1.5)"""
    if window_size != 11 or not size_average:
        raise NotImplementedError('d3h.ssim: window_size=11, size_average=True only')
    return _SSIMFn.apply(img1, img2)


# ---- the first k channels of a pixel-major tensor, with a one-pass backward ----------------------------------------------------------------
class _FirstChannelsFn(torch.autograd.Function):
    """x[..., :k] as a VIEW (no copy), whose gradient (g, zeros) is written by one kernel (csrc/image_ops.hip: channels_pad_kernel) -- autograd's own
    slice node fills a zero tensor of x's size and copies g into it: two passes, 59 us at 4 x 1024^2 x 6"""

    @staticmethod
    def forward(ctx, x, k):
        ctx.cout = int(x.shape[-1])
        return x[..., :k]

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        cin, cout = int(g.shape[-1]), ctx.cout
        out = torch.empty(*g.shape[:-1], cout, dtype=torch.float32, device=g.device)
        L.check(L.lib().d3h_channels_pad(L.ptr(g), L.i64(g.numel() // cin), L.i32(cin), L.i32(cout), L.ptr(out), L.stream()), 'channels_pad')
        return out, None


def first_channels(x, k):
    """x[..., :k] (a view); use where x is large and only these channels take a gradient"""
    if not (torch.is_grad_enabled() and x.requires_grad) or x.dtype != torch.float32 or k >= x.shape[-1]:
        return x[..., :k]
    return _FirstChannelsFn.apply(x, int(k))


# ---- composite of the layer buffers against their backgrounds -----------------------------------------------------------------
COMP_ZERO, COMP_IMAGE, COMP_CONST20, COMP_ALPHA = 0, 1, 2, 3


def _pix_view(t):
    """(tensor kept alive, pointer, floats between pixels) of a [..., c] slice of a pixel-major tensor, without copying"""
    t = t.float()
    c = t.shape[-1]
    st = t.stride()
    ok = st[-1] == 1 or c == 1
    ps = st[-2]
    n = 1
    for d in range(t.dim() - 2, -1, -1):       # every leading dim must continue the same pixel pitch
        ok = ok and (t.shape[d] == 1 or st[d] == ps * n)
        n *= t.shape[d]
    if not ok:
        t = t.contiguous()
        ps = c
    return t, ps


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rast, kinds, bgs, *srcs):
        import ctypes
        B, H, W = rast.shape[:3]
        dev = rast.device
        n = len(srcs)
        views = [_pix_view(s) for s in srcs]
        nch = [int(s.shape[-1]) for s in srcs]
        C = sum(1 if k == COMP_ALPHA else c + 1 for k, c in zip(kinds, nch))
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        bg_t = [None if b is None else b.float().contiguous() for b in bgs]
        P = ctypes.c_void_p * n
        I = ctypes.c_int * n
        addr = lambda t: None if t is None else t.data_ptr()
        rc = rast.contiguous()
        keep = [v[0] for v in views] + bg_t + [rc, out]
        L.check(L.lib().d3h_composite_fwd(L.i32(n), P(*[addr(v[0]) for v in views]), I(*[int(v[1]) for v in views]), I(*nch), I(*kinds),
                                          P(*[addr(b) for b in bg_t]), I(*[0 if b is None or b.shape[0] == 1 else 1 for b in bg_t]),
                                          L.ptr(rc), L.i32(B), L.i32(H), L.i32(W), L.ptr(out), L.stream()), 'composite_fwd')
        del keep
        ctx.save_for_backward(rc)
        ctx.meta = (kinds, nch, [s.shape for s in srcs])
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        rc, = ctx.saved_tensors
        kinds, nch, shapes = ctx.meta
        B, H, W = rc.shape[:3]
        n = len(nch)
        g = g.contiguous().float()
        ds = [torch.empty(B, H, W, c, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[3 + k] else None for k, c in enumerate(nch)]
        P = ctypes.c_void_p * n
        I = ctypes.c_int * n
        L.check(L.lib().d3h_composite_bwd(L.i32(n), P(*[None if d is None else d.data_ptr() for d in ds]), I(*nch), I(*kinds), L.ptr(rc),
                                          L.i32(B), L.i32(H), L.i32(W), L.ptr(g), L.stream()), 'composite_bwd')
        red = lambda d, shp: None if d is None else d.sum_to_size(shp)
        return (None, None, None) + tuple(red(d, shp) for d, shp in zip(ds, shapes))


def composite(rast, sources):
    """render/render.py:375-382,430-449 for all buffers at once.  sources: list of (values [B,H,W,c] (a strided slice is fine), kind,
    background) with kind COMP_ZERO / COMP_IMAGE (background [1|B,H,W,3], alpha 0) / COMP_CONST20 / COMP_ALPHA (c = 1: the value is
    the alpha, output one channel).  Returns [B,H,W,sum(c+1)] = lerp(background, [values, 1], coverage) per buffer, concatenated."""
    B, H, W = rast.shape[:3]
    srcs = [s.expand(B, H, W, s.shape[-1]) for s, _, _ in sources]
    return _CompositeFn.apply(rast, [k for _, k, _ in sources], [b for _, _, b in sources], *srcs)


def composite_antialias(rast, sources, pos, tri):
    """antialias(composite(rast, sources), rast, pos, tri) in ONE forward pass (csrc/raster.hip:aa_composite_fwd_kernel), for renders
    nobody differentiates: the dead buffers of a tick in its 'all' mode, the watertight validation render.  Bit-identical to the two
    separate ops; no autograd (call it under torch.no_grad())."""
    import ctypes
    from . import raster as _R
    assert not torch.is_grad_enabled() or not any(s.requires_grad for s, _, _ in sources), 'composite_antialias is forward-only'
    B, H, W = rast.shape[:3]
    dev = rast.device
    n = len(sources)
    srcs = [s.expand(B, H, W, s.shape[-1]) for s, _, _ in sources]
    kinds = [k for _, k, _ in sources]
    views = [_pix_view(s) for s in srcs]
    nch = [int(s.shape[-1]) for s in srcs]
    C = sum(1 if k == COMP_ALPHA else c + 1 for k, c in zip(kinds, nch))
    out = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
    bg_t = [None if b is None else b.float().contiguous() for _, _, b in sources]
    rc, pos_c, tri_c = rast.contiguous(), pos.contiguous().float(), tri.contiguous()
    flags = _R._edge_flags(pos_c, tri_c, B, H, W)
    P = ctypes.c_void_p * n
    I = ctypes.c_int * n
    addr = lambda t: None if t is None else t.data_ptr()
    keep = [v[0] for v in views] + bg_t
    L.check(L.lib().d3h_composite_antialias_fwd(L.i32(n), P(*[addr(v[0]) for v in views]), I(*[int(v[1]) for v in views]), I(*nch), I(*kinds),
                                                P(*[addr(b) for b in bg_t]), I(*[0 if b is None or b.shape[0] == 1 else 1 for b in bg_t]),
                                                L.ptr(rc), L.ptr(pos_c), L.i32(_R._bstride(pos_c)), L.ptr(tri_c), L.i32(tri_c.shape[0]), L.ptr(flags),
                                                L.i32(B), L.i32(H), L.i32(W), L.ptr(out), L.stream()), 'composite_antialias_fwd')
    del keep
    return out


class _CompositeAntialiasFn(torch.autograd.Function):
    """antialias(composite(rast, sources), rast, pos, tri) with its gradient, one kernel each way (csrc/raster.hip:aa_composite_fwd_kernel /
    aa_composite_bwd_kernel): the composited, pre-antialias image is never stored -- the backward re-evaluates it at the pair pixels.
    `use[k]`: source k is the first use[k] channels of the tensor passed in (0 = all of it); its gradient comes back at the tensor's full
    width, the unused channels zero-filled by the same kernel (the slice node + its pad pass of the separate path)."""

    @staticmethod
    def forward(ctx, rast, pos, tri, kinds, bgs, use, *srcs):
        import ctypes
        from . import raster as _R
        B, H, W = rast.shape[:3]
        n = len(srcs)
        views = [_pix_view(s if not u else s[..., :u]) for s, u in zip(srcs, use)]
        nch = [int(v[0].shape[-1]) for v in views]
        C = sum(1 if k == COMP_ALPHA else c + 1 for k, c in zip(kinds, nch))
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=rast.device)
        bg_t = [None if b is None else b.float().contiguous() for b in bgs]
        rc, pos_c, tri_c = rast.contiguous(), pos.contiguous().float(), tri.contiguous()
        flags = _R._edge_flags(pos_c, tri_c, B, H, W)
        P = ctypes.c_void_p * n
        I = ctypes.c_int * n
        addr = lambda t: None if t is None else t.data_ptr()
        keep = [v[0] for v in views] + bg_t
        L.check(L.lib().d3h_composite_antialias_fwd(L.i32(n), P(*[addr(v[0]) for v in views]), I(*[int(v[1]) for v in views]), I(*nch), I(*kinds),
                                                    P(*[addr(b) for b in bg_t]), I(*[0 if b is None or b.shape[0] == 1 else 1 for b in bg_t]),
                                                    L.ptr(rc), L.ptr(pos_c), L.i32(_R._bstride(pos_c)), L.ptr(tri_c), L.i32(tri_c.shape[0]), L.ptr(flags),
                                                    L.i32(B), L.i32(H), L.i32(W), L.ptr(out), L.stream()), 'composite_antialias_fwd')
        del keep
        ctx.save_for_backward(rc, pos_c, tri_c, flags, *[v[0] for v in views], *[b for b in bg_t if b is not None])
        ctx.meta = (kinds, nch, [s.shape for s in srcs], [int(v[1]) for v in views], [b is not None for b in bg_t])
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import raster as _R
        kinds, nch, shapes, strides, has_bg = ctx.meta
        n = len(nch)
        saved = ctx.saved_tensors
        rc, pos_c, tri_c, flags = saved[:4]
        views = saved[4:4 + n]
        bgs_it = iter(saved[4 + n:])
        bg_t = [next(bgs_it) if h else None for h in has_bg]
        B, H, W = rc.shape[:3]
        g = g.contiguous().float()
        dch = [int(shp[-1]) for shp in shapes]                      # the gradient comes back at the width of the tensor passed in
        ds = [torch.empty(B, H, W, c, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[6 + k] else None for k, c in enumerate(dch)]
        d_pos = L.zeros_like(pos_c) if ctx.needs_input_grad[1] else None
        P = ctypes.c_void_p * n
        I = ctypes.c_int * n
        addr = lambda t: None if t is None else t.data_ptr()
        L.check(L.lib().d3h_composite_antialias_bwd(L.i32(n), P(*[addr(v) for v in views]), P(*[addr(d) for d in ds]), I(*dch), I(*strides), I(*nch),
                                                    I(*kinds), P(*[addr(b) for b in bg_t]), I(*[0 if b is None or b.shape[0] == 1 else 1 for b in bg_t]),
                                                    L.ptr(rc), L.ptr(pos_c), L.i32(_R._bstride(pos_c)), L.ptr(tri_c), L.i32(tri_c.shape[0]), L.ptr(flags),
                                                    L.i32(B), L.i32(H), L.i32(W), L.ptr(g), L.ptr(d_pos), L.stream()), 'composite_antialias_bwd')
        red = lambda d, shp: None if d is None else d.sum_to_size(shp)
        return (None, d_pos, None, None, None, None) + tuple(red(d, shp) for d, shp in zip(ds, shapes))


def composite_antialias_grad(rast, sources, pos, tri):
    """antialias(composite(rast, sources), rast, pos, tri), differentiable in the sources and in `pos`, as one kernel forward and one
    backward.  Values bit-identical to the two separate ops, source gradients too; d(pos) up to the order of its float atomics.
    A source may be given as (tensor, kind, background, k): the buffer is tensor[..., :k] (see _CompositeAntialiasFn)."""
    B, H, W = rast.shape[:3]
    srcs = [s[0].expand(B, H, W, s[0].shape[-1]) for s in sources]
    use = [int(s[3]) if len(s) > 3 and s[3] and s[3] < s[0].shape[-1] else 0 for s in sources]
    return _CompositeAntialiasFn.apply(rast, pos, tri, [s[1] for s in sources], [s[2] for s in sources], use, *srcs)


# ---- fused per-pixel loss stack of tick_init / tick_split -------------------------------------------------------------------
SSIM_OCC = os.environ.get('D3H_SSIM_OCC', '1') != '0'
PIXEL_LOSS_KEYS = ('mask_mse', 'img', 'msdf_pos_l1', 'msdf_neg_l1', 'normal_mse', 'normal_cos', 'kd_grad', 'ks_grad', 'normal_grad', 'ssim')


class _PixelLossesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stacked, cref, nref, cs, cg, cm, ckg, csg, cng, loss, tonemap, want_ssim, masked_prep):
        st = stacked.contiguous().float()
        B, H, W, C = st.shape
        cr = cref.float().expand(B, H, W, 4).contiguous()
        nr = None
        if nref is not None:
            nr = nref.float().expand(B, H, W, nref.shape[-1])
            nr = nr if nr.is_contiguous() else nr.contiguous()
        dev = st.device
        sums = torch.empty(10, dtype=torch.float32, device=dev)
        npix = B * H * W
        sa = sb = None
        if want_ssim:
            sa = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
            sb = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        lib = L.lib()
        # masked_prep: None = no extra output; () = the masked colour image shaded.rgb * ref.a; (shift[3], scale[3]) = that image mapped
        # to the LPIPS trunk input ((2 x - 1) - shift) / scale -- a second, differentiable output [B,H,W,3]
        masked = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev) if masked_prep is not None else None
        # the occupancy cells of the two SSIM operands (csrc/image_ops.hip: SsimOcc): both are zero wherever the target's alpha is, and the
        # SSIM passes skip the bands that see nothing else -- same results; D3H_SSIM_OCC=0: every band is computed
        occ = L.zeros((B, -(-H // 32), -(-W // 64)), torch.int32, dev) if (want_ssim and SSIM_OCC) else None
        prep = (ctypes.c_float * 6)(*[float(v) for v in (list(masked_prep[0]) + list(masked_prep[1]))]) if masked_prep else None
        L.check(lib.d3h_pixel_losses_fwd(L.ptr(st), L.i32(C), L.i32(cs), L.i32(cg), L.i32(cm), L.i32(ckg), L.i32(csg), L.i32(cng), L.ptr(cr), L.ptr(nr),
                                         L.i32(0 if nr is None else nr.shape[-1]), L.i32(B), L.i32(H), L.i32(W), L.i32(loss), L.i32(tonemap),
                                         L.ptr(sums), L.ptr(sa), L.ptr(sb), L.ptr(masked), prep, L.ptr(occ), L.stream()), 'pixel_losses_fwd')
        gmom = None
        need = stacked.requires_grad
        if want_ssim:
            N = 3 * B
            tmp = None
            gmom = torch.empty(5 * N * H * W, dtype=torch.float32, device=dev) if need else None
            L.check(lib.d3h_ssim_fwd(L.ptr(sa), L.ptr(sb), L.i32(N), L.i32(H), L.i32(W), L.ptr(tmp), L.ptr(gmom), L.i32(0), L.ptr(sums[9:]), L.ptr(occ),
                                     L.i32(3), L.stream()), 'ssim_fwd')          # (0: the masked target is a constant)
        else:
            sums[9:].zero_()
        scale = _const([1.0 / npix] * 4 + [1.0 / (3 * npix), 1.0 / npix, 1.0 / npix, 1.0 / (3 * npix), 1.0 / (3 * npix), 1.0 / (3 * npix)], dev) \
            if npix else torch.zeros(10, device=dev)
        ctx.cfg = (B, H, W, C, cs, cg, cm, ckg, csg, cng, loss, tonemap, want_ssim, prep)
        ctx.save_for_backward(st, cr, nr, sa, sb, gmom, scale, occ)
        if masked is None:
            return sums * scale
        return sums * scale, masked

    @staticmethod
    def backward(ctx, g, g_masked=None):
        st, cr, nr, sa, sb, gmom, scale, occ = ctx.saved_tensors
        B, H, W, C, cs, cg, cm, ckg, csg, cng, loss, tonemap, want_ssim, prep = ctx.cfg
        lib = L.lib()
        gs = (g.float() * scale).contiguous() if g is not None else torch.zeros(10, dtype=torch.float32, device=st.device)
        gm = g_masked.contiguous().float() if g_masked is not None else None
        d_a = None
        if want_ssim:
            N = 3 * B
            tmp = None
            d_a = torch.empty_like(sa)
            L.check(lib.d3h_ssim_bwd(L.ptr(sa), L.ptr(sb), L.i32(N), L.i32(H), L.i32(W), L.ptr(gmom), L.i32(0), L.ptr(tmp), L.ptr(gs[9:]), L.f32(1.0),
                                     L.ptr(d_a), L.ptr(None), L.ptr(occ), L.i32(3), L.stream()), 'ssim_bwd')
        d_st = torch.empty_like(st)
        L.check(lib.d3h_pixel_losses_bwd(L.ptr(st), L.i32(C), L.i32(cs), L.i32(cg), L.i32(cm), L.i32(ckg), L.i32(csg), L.i32(cng), L.ptr(cr), L.ptr(nr),
                                         L.i32(0 if nr is None else nr.shape[-1]), L.i32(B), L.i32(H), L.i32(W), L.i32(loss), L.i32(tonemap),
                                         L.ptr(gs), L.ptr(d_a), L.ptr(gm), prep if gm is not None else None, L.ptr(d_st), L.stream()), 'pixel_losses_bwd')
        return (d_st,) + (None,) * 12


def pixel_losses(stacked, layout, color_ref, normal_ref=None, image_loss_spec=None, want_ssim=False, masked_prep=None):
    """The per-pixel losses both tick_init and tick_split evaluate (hmsdf.py:835-839,895-898 / 969-975,1064-1068) in one pass over
    render_mesh's channel-concatenated output.  stacked: [B,H,W,C]; layout: {buffer: (first channel, channels)}; returns a dict of
    MEANS: mask_mse = mse(shaded.a, ref.a); img = image_loss(shaded.rgb*ref.a, ref.rgb*ref.a) for image_loss_spec = (loss,
    tonemapper) (0 if None); msdf_pos_l1 / msdf_neg_l1 = the two L1 terms on msdf_image; normal_mse / normal_cos = mse and mean
    cosine between normalize(geometric_normal)*(1,-1,-1) and normalize(normal_ref); kd_grad / ks_grad / normal_grad = the three means of
    regularizer.material_smoothness_grad (before their lambdas; 0 for absent buffers); ssim = ssim_loss.ssim of the masked images."""
    ch = lambda k: layout[k][0] if k in layout else -1
    loss, tone = (-1, 0) if image_loss_spec is None else (_LOSS[image_loss_spec[0]], _TONE[image_loss_spec[1]])
    v = _PixelLossesFn.apply(stacked, color_ref, normal_ref, ch('shaded'), ch('geometric_normal') if normal_ref is not None else -1,
                             ch('msdf_image'), ch('kd_grad'), ch('ks_grad'), ch('normal_grad'), loss, tone, bool(want_ssim), masked_prep)
    masked = None
    if masked_prep is not None:          # `masked` [B,H,W,3]: shaded.rgb * ref.a (optionally as the LPIPS trunk input), differentiable
        v, masked = v
    d = dict(zip(PIXEL_LOSS_KEYS, v.unbind(0)))
    d['vec'] = v
    d['masked'] = masked
    return d


# ---- mask and image terms of tick_seq ------------------------------------------------------------------------------------------
SEQ_LOSS_KEYS = ('all_msk', 'cloth_msk', 'body_msk', 'all_img', 'cloth_img', 'body_img')


class _SeqLossesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stacked, label, gt_all, gt_cloth, gt_body, cs, ca, loss, tonemap):
        st = stacked.contiguous().float()
        B, H, W, C = st.shape
        npix = B * H * W
        c4 = lambda t: t.float().expand(B, H, W, 4).contiguous()
        lab = label.float().expand(B, H, W).contiguous()
        ga, gc, gb = c4(gt_all), c4(gt_cloth), c4(gt_body)
        sums = torch.empty(6, dtype=torch.float32, device=st.device)
        L.check(L.lib().d3h_seq_losses_fwd(L.ptr(st), L.i32(C), L.i32(cs), L.i32(ca), L.ptr(lab), L.ptr(ga), L.ptr(gc), L.ptr(gb), L.i64(npix), L.i32(loss),
                                           L.i32(tonemap), L.ptr(sums), L.stream()), 'seq_losses_fwd')
        ctx.save_for_backward(st, lab, ga, gc, gb)
        ctx.cfg = (C, cs, ca, loss, tonemap, npix)
        return sums * (1.0 / max(npix, 1))

    @staticmethod
    def backward(ctx, g):
        st, lab, ga, gc, gb = ctx.saved_tensors
        C, cs, ca, loss, tonemap, npix = ctx.cfg
        gs = (g.float() * (1.0 / max(npix, 1))).contiguous()
        d_st = torch.empty_like(st)
        L.check(L.lib().d3h_seq_losses_bwd(L.ptr(st), L.i32(C), L.i32(cs), L.i32(ca), L.ptr(lab), L.ptr(ga), L.ptr(gc), L.ptr(gb), L.i64(npix), L.i32(loss),
                                           L.i32(tonemap), L.ptr(gs), L.ptr(d_st), L.stream()), 'seq_losses_bwd')
        return (d_st,) + (None,) * 8


def seq_losses(stacked, layout, label, gt_all, gt_cloth, gt_body, image_loss_spec):
    """The three mask MSEs and the three image losses of tick_seq (hmsdf.py:787-797,1110-1123) as MEANS, in SEQ_LOSS_KEYS order: masks
    alpha, label * alpha, (1 - label) * alpha from the coverage channel of `geometric_normal`; image losses ru.image_loss(shaded.rgb * mask,
    gt.rgb) for image_loss_spec = (loss, tonemapper).  stacked / layout: render_mesh's '_stacked' / '_layout'; label [B,H,W] (no gradient)."""
    cs = layout['shaded'][0]
    ca = layout['geometric_normal'][0] + layout['geometric_normal'][1] - 1
    return _SeqLossesFn.apply(stacked, label, gt_all, gt_cloth, gt_body, cs, ca, _LOSS[image_loss_spec[0]], _TONE[image_loss_spec[1]])


# ---- SDF edge regulariser ----------------------------------------------------------------------------------
class _SdfRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, edges32, marks):
        s = sdf.reshape(-1).contiguous().float()
        sums = torch.empty(2, dtype=torch.float32, device=s.device)
        L.check(L.lib().d3h_sdf_reg_fwd(L.ptr(s), L.ptr(edges32), L.i32(edges32.shape[0]), L.ptr(sums), L.ptr(marks), L.stream()), 'sdf_reg_fwd')
        ctx.save_for_backward(s, edges32, sums)
        ctx.shape = sdf.shape
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        s, edges32, sums = ctx.saved_tensors
        d = L.zeros_like(s)
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_sdf_reg_bwd(L.ptr(s), L.ptr(edges32), L.i32(edges32.shape[0]), L.ptr(sums), L.ptr(gs), L.ptr(d), L.stream()),
                'sdf_reg_bwd')
        return d.reshape(ctx.shape), None, None


def sdf_reg_loss(sdf, edges32, marks=None):
    """geometry/hmsdf.py:162-170 compute_sdf_reg_loss (edges32: int32 [N_e,2] == all_edges).  marks (optional): float32 [sdf.numel()], zero on
    entry; receives 1 at both ends of every sign-changing edge (what d3h.sdf_mlp.prepare_backward wants)"""
    return _SdfRegFn.apply(sdf, edges32.contiguous(), marks)


# ---- xfm_points -------------------------------------------------------------------------------------------
class _XfmPointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, mtx, w):
        p = pts.contiguous().float()
        M = mtx.detach().contiguous().float()
        nb, n = M.shape[0], p.shape[1]
        out = torch.empty(nb, n, 4, dtype=torch.float32, device=p.device)
        L.check(L.lib().d3h_xfm_points_fwd(L.ptr(p), L.i32(p.shape[0]), L.ptr(M), L.i32(nb), L.i32(n), L.f32(w), L.ptr(out), L.stream()), 'xfm_points_fwd')
        ctx.save_for_backward(M)
        ctx.pshape = p.shape
        return out

    @staticmethod
    def backward(ctx, g):
        M, = ctx.saved_tensors
        nbp, n, _ = ctx.pshape
        d = torch.empty(ctx.pshape, dtype=torch.float32, device=g.device)
        L.check(L.lib().d3h_xfm_points_bwd(L.ptr(g.contiguous().float()), L.i32(nbp), L.ptr(M), L.i32(M.shape[0]), L.i32(n), L.ptr(d), L.stream()),
                'xfm_points_bwd')
        return d, None, None


def xfm_points(points, matrix, w=1.0):
    """[B or 1,V,3] x [B,4,4] -> [B,V,4]: M [p; w] per point (render/renderutils/ops.py:518-537).  The matrix is treated as a constant
    (callers with a trainable matrix use the matmul formulation)."""
    return _XfmPointsFn.apply(points, matrix, float(w))


class _LpipsHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f0, n1, w):
        B, C, H, W = f0.shape
        # the feature maps of a convolution stack fed with an NHWC-strided image are channels-last: read them as they are (a
        # .contiguous() here would be a transposing copy of 16-66 MB per layer, and another one for the gradient on the way back)
        cl = C % 64 == 0 and f0.is_contiguous(memory_format=torch.channels_last) and not f0.is_contiguous()
        fmt = torch.channels_last if cl else torch.contiguous_format
        f0c, n1c = f0.contiguous(memory_format=fmt).float(), n1.contiguous(memory_format=fmt).float()
        out = torch.empty(B, dtype=torch.float32, device=f0c.device)
        L.check(L.lib().d3h_lpips_head_fwd(L.ptr_any(f0c), L.ptr_any(n1c), L.ptr(w), L.i32(B), L.i32(C), L.i32(H * W), L.i32(int(cl)), L.ptr(out),
                                           L.stream()), 'lpips_head_fwd')
        ctx.save_for_backward(f0c, n1c, w)
        ctx.cl = cl
        return out

    @staticmethod
    def backward(ctx, g):
        f0c, n1c, w = ctx.saved_tensors
        B, C, H, W = f0c.shape
        d = torch.empty_like(f0c)                       # preserves the memory format
        L.check(L.lib().d3h_lpips_head_bwd(L.ptr_any(f0c), L.ptr_any(n1c), L.ptr(w), L.i32(B), L.i32(C), L.i32(H * W), L.i32(int(ctx.cl)),
                                           L.ptr(g.contiguous().float()), L.ptr_any(d), L.stream()), 'lpips_head_bwd')
        return d, None, None


def lpips_head(f0, n1, w):
    """one LPIPS layer: f0 [B,C,H,W] features of the prediction, n1 the unit-normalised features of the reference (constant), w [C] the
    layer's linear weights (constant) -> [B] = spatial mean of sum_c w_c (f0 / (|f0| + 1e-10) - n1)^2   (csrc/lpips_head.hip)"""
    return _LpipsHeadFn.apply(f0, n1.detach(), w.detach().reshape(-1).contiguous().float())


# ---- smoothness buffers of shade() (kd, kd_grad, ks_grad, normal_grad) in one pass --------------------------------------------
class _MaterialGradsFn(torch.autograd.Function):
    """render.py:72-74,88-91,104-105; csrc/material_grads.hip.  Outputs are separate contiguous tensors (their gradients arrive as such: no
    slice nodes); every input gradient is written completely by one backward launch."""

    @staticmethod
    def forward(ctx, tex, texj, nrm, nrmj, mask, mask_tap, want_kd=True):
        lib = L.lib()
        shp = tex.shape[:-1]
        c = lambda t: None if t is None else t.contiguous().float()
        tex, texj, nrm, nrmj, mask, mask_tap = c(tex), c(texj), c(nrm), c(nrmj), c(mask), c(mask_tap)
        n = tex.numel() // 6
        new = lambda: torch.empty(*shp, 3, dtype=torch.float32, device=tex.device)
        kd = new() if want_kd else None
        kdg, ksg = (new(), new()) if texj is not None else (None, None)
        ng = new() if nrm is not None else None
        L.check(lib.d3h_material_grads_fwd(L.ptr(tex), L.ptr(texj), L.ptr(nrm), L.ptr(nrmj), L.ptr(mask), L.ptr(mask_tap), L.i64(n), L.ptr(kd),
                                           L.ptr(kdg), L.ptr(ksg), L.ptr(ng), L.stream()), 'material_grads_fwd')
        ctx.save_for_backward(tex, texj, nrm, nrmj, mask, mask_tap)
        ctx.shp = shp
        return kd, kdg, ksg, ng

    @staticmethod
    def backward(ctx, g_kd, g_kdg, g_ksg, g_ng):
        tex, texj, nrm, nrmj, mask, mask_tap = ctx.saved_tensors
        lib = L.lib()
        shp = ctx.shp
        n = tex.numel() // 6
        c = lambda t: None if t is None else t.contiguous().float()
        g_kd, g_kdg, g_ksg, g_ng = c(g_kd), c(g_kdg), c(g_ksg), c(g_ng)
        need_tex = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        need_nrm = nrm is not None and (ctx.needs_input_grad[2] or ctx.needs_input_grad[3])
        d_tex = torch.empty(*shp, 6, dtype=torch.float32, device=tex.device) if need_tex else None
        d_texj = torch.empty_like(d_tex) if (need_tex and texj is not None) else None
        d_nrm = torch.empty(*shp, 3, dtype=torch.float32, device=tex.device) if need_nrm else None
        d_nrmj = torch.empty_like(d_nrm) if need_nrm else None
        if need_tex or need_nrm:
            L.check(lib.d3h_material_grads_bwd(L.ptr(tex), L.ptr(texj), L.ptr(nrm), L.ptr(nrmj), L.ptr(mask), L.ptr(mask_tap), L.i64(n), L.ptr(g_kd),
                                               L.ptr(g_kdg), L.ptr(g_ksg), L.ptr(g_ng), L.ptr(d_tex), L.ptr(d_texj), L.ptr(d_nrm), L.ptr(d_nrmj),
                                               L.stream()), 'material_grads_bwd')
        return d_tex, d_texj, d_nrm, d_nrmj, None, None, None


def material_grads(all_tex, all_tex_jitter=None, gb_normal=None, nrm_jitter=None, mask=None, mask_tap=None, want_kd=True):
    """-> (kd, kd_grad, ks_grad, normal_grad); kd_grad / ks_grad are None without all_tex_jitter, normal_grad without the normal inputs, kd
    without want_kd"""
    return _MaterialGradsFn.apply(all_tex, all_tex_jitter, gb_normal, nrm_jitter, mask, mask_tap, want_kd)
