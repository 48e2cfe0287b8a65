"""Host side of the fused SDF-query kernels (csrc/sdf_mlp.hip): weight packing + forward (+ backward).

The network shape is the reference's fixed working point (train.py:1618-1621): MLP(n_freq=6, d_hidden=256,
n_hidden=6, skip_in=[3]) from geometry/mlp.py:10-32; any other shape raises (no silent fallback).
"""
import os

import torch
from d3h._lib import cur_stream as _cur_stream

from . import _lib as L
from . import gradarena as _GA

HIDDEN_KEYS = (2, 4, 6, 10, 12)
SPARSE_BACKWARD = True      # skip 16-point tiles whose upstream gradient is identically zero (exact)
# The training sweep runs WITHOUT the activation save and the backward recomputes the activations of the tiles it visits (the active ~15 % of a
# grid sweep) with the same kernel -- bit-identical values -- instead of storing 1.88 GB per 262 144 points of which ~15 % was read back.
# Only with the bf16 x 3 kernels (the recompute pass is one); D3H_SDF_RECOMPUTE=0 restores the store.
RECOMPUTE = os.environ.get('D3H_SDF_RECOMPUTE', '1') != '0' 
# The forward and tangent sweeps run their GEMMs on the bf16 matrix pipe with every fp32 operand split into three bf16 numbers
# (csrc/sdf_mlp_x3.h: fp32-level accuracy at 3/8 of the exact-f32 MFMA's pipe time); D3H_SDF_X3=0 selects the exact-f32 MFMA kernels.
X3 = os.environ.get('D3H_SDF_X3', '1') != '0'
# The forward-type sweeps (the grid sweep, the eikonal forward, the recompute of the sparse backward) split their operands into TWO fp16 planes
# instead of three bf16 ones: three matrix-core products per block instead of six at the same fp32-level accuracy (csrc/sdf_mlp_x3.h "h2").
# D3H_SDF_H2=0 keeps them on the bf16 x 3 split.  The tangent / data-backward / weight-gradient sweeps are bf16 x 3 either way.
H2 = os.environ.get('D3H_SDF_H2', '1') != '0'
H2_BWD = os.environ.get('D3H_SDF_H2_BWD', '1') != '0'      # '0': the data-backward sweeps stay on bf16 x 3 (A/B)
H2_JVP = os.environ.get('D3H_SDF_H2_JVP', '1') != '0'      # '0': the tangent sweep of the eikonal term stays on bf16 x 3 (A/B)
TIMING = None      # bench.py sets this to a list: (start_event, end_event, n_points) per forward launch, on the launch stream


def check_shape(sd, prefix='net.'):
    want = {0: (256, 39), 2: (256, 256), 4: (256, 256), 6: (256, 256), 8: (256, 295), 10: (256, 256), 12: (256, 256),
            14: (1, 256)}
    for i, shp in want.items():
        w = sd.get(f'{prefix}{i}.weight')
        if w is None or tuple(w.shape) != shp:
            raise RuntimeError(f'd3h.sdf_mlp: unsupported MLP shape at {prefix}{i} '
                               f'({None if w is None else tuple(w.shape)}; kernel is built for n_freq=6,d_hidden=256,'
                               f'n_hidden=6,skip_in=[3])')


def pack_weights(sd, prefix='net.', out=None):
    """state_dict-like {net.i.weight, net.i.bias} -> packed fragment-order buffer (see csrc/sdf_mlp_layout.h)."""
    check_shape(sd, prefix)
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    bh = torch.stack([g(f'{i}.bias') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), g('0.bias'), wh, bh, g('8.weight'), g('8.bias'), g('14.weight'), g('14.bias')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpack_floats(), dtype=torch.float32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack')
    return out


def forward(x, wpack, deform=None, disp=0.0, save=False, want_xdef=False, max_cus=0, wp3=None):
    """sdf[n] (and optionally the saved activations / deformed points) for points x[n,3].  max_cus: a launch of fewer than 1024 point
    tiles uses at most this many CUs (0 = the chip), so that another stream's kernels find free ones next to it."""
    lib = L.lib()
    x = x.contiguous().float()
    n = x.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=x.device)
    act = torch.empty(lib.d3h_sdf_mlp_act_floats(n), dtype=torch.float32, device=x.device) if save else None
    xdef = torch.empty(n, 3, dtype=torch.float32, device=x.device) if want_xdef else None
    d = deform.contiguous().float() if deform is not None else None
    ev = None
    if TIMING is not None and x.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if wp3 is not None and getattr(wp3, 'd3h_planes', 3) == 2:       # a pack of pack_weights_h2: the fp16 x 2 sweep
        L.check(lib.d3h_sdf_mlp_fwd_h2(L.ptr(x), L.ptr(d), L.f32(disp), L.ptr(wp3), L.ptr(sdf), L.ptr(xdef), L.ptr(act), L.i64(n),
                                       L.i32(max_cus), L.stream()), 'sdf_mlp_fwd_h2')
    elif wp3 is not None:       # same outputs on the bf16 matrix pipe (wpack3 of the same weights: pack_weights3 / PackedWeights.wp3)
        L.check(lib.d3h_sdf_mlp_fwd_x3(L.ptr(x), L.ptr(d), L.f32(disp), L.ptr(wp3), L.ptr(sdf), L.ptr(xdef), L.ptr(act), L.i64(n),
                                       L.i32(max_cus), L.stream()), 'sdf_mlp_fwd_x3')
    else:
        L.check(lib.d3h_sdf_mlp_fwd(L.ptr(x), L.ptr(d), L.f32(disp), L.ptr(wpack), L.ptr(sdf), L.ptr(xdef), L.ptr(act), L.i64(n),
                                    L.i32(max_cus), L.stream()), 'sdf_mlp_fwd')
    if ev is not None:
        ev[1].record()
        TIMING.append((ev[0], ev[1], n))
    if save or want_xdef:
        return sdf, act, xdef
    return sdf


def pack_weights3(sd, prefix='net.', out=None):
    """the bf16 x 3 pack of the same weights for forward(..., wp3=) (csrc/sdf_mlp_x3.h)"""
    check_shape(sd, prefix)
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    bh = torch.stack([g(f'{i}.bias') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), g('0.bias'), wh, bh, g('8.weight'), g('8.bias'), g('14.weight'), g('14.bias')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpack3_dwords(), dtype=torch.int32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack3(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack3')
    return out


def pack_weights_h2(sd, prefix='net.', out=None):
    """the fp16 x 2 pack of the same weights for forward(..., wp3=<this>) (csrc/sdf_mlp_x3.h "h2"); the result is tagged `d3h_planes = 2`"""
    check_shape(sd, prefix)
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    bh = torch.stack([g(f'{i}.bias') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), g('0.bias'), wh, bh, g('8.weight'), g('8.bias'), g('14.weight'), g('14.bias')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpackh2_dwords(), dtype=torch.int32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack_h2(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack_h2')
    out.d3h_planes = 2
    return out


def pack_weights_t(sd, prefix='net.', out=None):
    """transposed fragment-order pack for the backward-data kernel"""
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), wh, g('8.weight')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpackt_floats(), dtype=torch.float32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack_t(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack_t')
    return out


# ---- one flat parameter vector --------------------------------------------------------------------------------------------------
# The 16 parameter tensors of the network are concatenated ONCE per iteration into a flat vector in "arena order" (the order the
# gradient kernels write: W0, b0, the five 256x256 hidden weights stacked, their biases stacked, W8 (the 295-wide skip layer), b8, W14,
# b14).  The sweep and the eikonal term both consume that vector, so the autograd engine sums their two gradients with ONE 1.66 MB add
# (it used to be 16 small adds, 5 us apart, on the launch-bound tail of the backward), the weight packs read views of it (no
# torch.stack), and the backward hands the 16 parameter gradients back as views of one arena (no copies).
ARENA_SIZES = [256 * 39, 256, 5 * 65536, 5 * 256, 256 * 295, 256, 256, 1]
ARENA_FLOATS = sum(ARENA_SIZES)
_ARENA_PERM = [0, 1, 2, 4, 6, 10, 12, 3, 5, 7, 11, 13, 8, 9, 14, 15]          # positions in _PARAM_ORDER, in arena order
_ARENA_SHAPES = [(256, 39), (256,)] + [(256, 256)] * 5 + [(256,)] * 5 + [(256, 295), (256,), (1, 256), (1,)]


class _FlatParams(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        for k, shp in zip(_ARENA_PERM, _ARENA_SHAPES):
            if tuple(params[k].shape) != shp:
                raise RuntimeError(f'd3h.sdf_mlp: unsupported MLP shape {tuple(params[k].shape)} for net.{_PARAM_ORDER[k]} (kernel is built for '
                                   f'n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3])')
        ctx.leaves = [params[k] for k in _ARENA_PERM]
        return torch.cat([params[k].detach().reshape(-1).float() for k in _ARENA_PERM])

    @staticmethod
    def backward(ctx, g):
        # frame-parallel step: the 16 gradients live in the step's all-reduce arena (d3h.gradarena) -- one 1.66 MB copy of the summed
        # vector into its block, the views below are then slices of the bucket the collective runs on
        blk = _GA.block_for(ctx.leaves)
        ctx.leaves = None
        if blk is not None:
            blk.copy_(g)
            g = blk
        out = [None] * 16
        o = 0
        for k, shp in zip(_ARENA_PERM, _ARENA_SHAPES):
            n = 1
            for d in shp:
                n *= d
            out[k] = g[o:o + n].view(shp)
            o += n
        return tuple(out)


def arena_views(flat):
    """(W0, b0, Wh[5,256,256], bh[5,256], W8, b8, W14, b14) views of a flat arena-order vector"""
    a = torch.split(flat, ARENA_SIZES)
    return a[0].view(256, 39), a[1], a[2].view(5, 256, 256), a[3].view(5, 256), a[4].view(256, 295), a[5], a[6].view(1, 256), a[7]


def pack_weights_t_h2(sd, prefix='net.', out=None):
    """the fp16 x 2 transposed pack for the data-backward sweeps (csrc/sdf_mlp_x3.h "h2"); tagged `d3h_planes = 2`"""
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), wh, g('8.weight')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpackth2_dwords(), dtype=torch.int32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack_t_h2(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack_t_h2')
    out.d3h_planes = 2
    return out


def _planes(t):
    return 0 if t is None else int(getattr(t, 'd3h_planes', 3))


def pack_weights_t3(sd, prefix='net.', out=None):
    """the bf16 x 3 transposed pack for the data-backward sweeps (csrc/sdf_mlp_x3.h)"""
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), wh, g('8.weight')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpackt3_dwords(), dtype=torch.int32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack_t3(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack_t3')
    return out


class PackedWeights:
    """the flat parameter vector of one parameter state plus both fragment-order packs of it (forward + transposed), built once per sweep
    and shared by the sweep, its backward and the eikonal term of the same iteration.  `valid_for` compares the autograd version
    counters: any optimiser step invalidates it."""

    def __init__(self, params):
        self.key = tuple((p.data_ptr(), p._version) for p in params)
        self.flat = _FlatParams.apply(*params)                  # differentiable: both consumers' gradients meet here
        lib = L.lib()
        w0, b0, wh, bh, w8, b8, w14, b14 = arena_views(self.flat.detach())
        self.wp = self.wpt = self.wp3 = self.wpt3 = self.wph = None
        dbg = os.environ.get('D3H_X3_DEBUG_PARTS')      # diagnostic: e.g. "fwd,eik" keeps only those sweeps on the bf16 pipe (both packs built)
        if X3 and dbg is not None:
            self.wp = torch.empty(lib.d3h_sdf_mlp_wpack_floats(), dtype=torch.float32, device=w0.device)
            L.check(lib.d3h_sdf_mlp_pack(L.ptr(w0), L.ptr(b0), L.ptr(wh), L.ptr(bh), L.ptr(w8), L.ptr(b8), L.ptr(w14), L.ptr(b14), L.ptr(self.wp),
                                         L.stream()), 'sdf_mlp_pack')
            self.wpt = torch.empty(lib.d3h_sdf_mlp_wpackt_floats(), dtype=torch.float32, device=w0.device)
            L.check(lib.d3h_sdf_mlp_pack_t(L.ptr(w0), L.ptr(wh), L.ptr(w8), L.ptr(self.wpt), L.stream()), 'sdf_mlp_pack_t')
        if X3:       # the split-plane packs carry everything the sweeps need (biases and head included): the f32 packs are not built
            if not (H2 and H2_JVP) or dbg is not None:          # (with every forward-type sweep on the fp16 x 2 pack nobody reads the bf16 x 3 one)
                self.wp3 = torch.empty(lib.d3h_sdf_mlp_wpack3_dwords(), dtype=torch.int32, device=w0.device)
                L.check(lib.d3h_sdf_mlp_pack3(L.ptr(w0), L.ptr(b0), L.ptr(wh), L.ptr(bh), L.ptr(w8), L.ptr(b8), L.ptr(w14), L.ptr(b14), L.ptr(self.wp3),
                                              L.stream()), 'sdf_mlp_pack3')
            if H2 and H2_BWD:      # the data-backward sweeps on the fp16 x 2 split too (operands scaled per launch, csrc/sdf_mlp_x3.h: h2_grad_scale)
                self.wpt3 = torch.empty(lib.d3h_sdf_mlp_wpackth2_dwords(), dtype=torch.int32, device=w0.device)
                L.check(lib.d3h_sdf_mlp_pack_t_h2(L.ptr(w0), L.ptr(wh), L.ptr(w8), L.ptr(self.wpt3), L.stream()), 'sdf_mlp_pack_t_h2')
                self.wpt3.d3h_planes = 2
            else:
                self.wpt3 = torch.empty(lib.d3h_sdf_mlp_wpackt3_dwords(), dtype=torch.int32, device=w0.device)
                L.check(lib.d3h_sdf_mlp_pack_t3(L.ptr(w0), L.ptr(wh), L.ptr(w8), L.ptr(self.wpt3), L.stream()), 'sdf_mlp_pack_t3')
            if H2:      # the forward-type sweeps read this one (fp16 x 2); wp3 stays for the tangent sweep
                self.wph = torch.empty(lib.d3h_sdf_mlp_wpackh2_dwords(), dtype=torch.int32, device=w0.device)
                L.check(lib.d3h_sdf_mlp_pack_h2(L.ptr(w0), L.ptr(b0), L.ptr(wh), L.ptr(bh), L.ptr(w8), L.ptr(b8), L.ptr(w14), L.ptr(b14), L.ptr(self.wph),
                                                L.stream()), 'sdf_mlp_pack_h2')
                self.wph.d3h_planes = 2
        else:
            self.wp = torch.empty(lib.d3h_sdf_mlp_wpack_floats(), dtype=torch.float32, device=w0.device)
            L.check(lib.d3h_sdf_mlp_pack(L.ptr(w0), L.ptr(b0), L.ptr(wh), L.ptr(bh), L.ptr(w8), L.ptr(b8), L.ptr(w14), L.ptr(b14), L.ptr(self.wp),
                                         L.stream()), 'sdf_mlp_pack')
            self.wpt = torch.empty(lib.d3h_sdf_mlp_wpackt_floats(), dtype=torch.float32, device=w0.device)
            L.check(lib.d3h_sdf_mlp_pack_t(L.ptr(w0), L.ptr(wh), L.ptr(w8), L.ptr(self.wpt), L.stream()), 'sdf_mlp_pack_t')
        self.w14 = w14

    def valid_for(self, params):
        return self.key == tuple((p.data_ptr(), p._version) for p in params)

    @property
    def wpf(self):
        """the pack of the forward-type sweeps: fp16 x 2 when built (H2), else bf16 x 3 (None with D3H_SDF_X3=0)"""
        return self.wph if self.wph is not None else self.wp3


def _part(t, name):
    """diagnostic switch D3H_X3_DEBUG_PARTS (see PackedWeights): the bf16-plane pack, or None (= exact-f32 kernels) for sweeps not listed"""
    dbg = os.environ.get('D3H_X3_DEBUG_PARTS')
    return t if (dbg is None or name in dbg.split(',')) else None


def _packs(pack, params):
    if pack is not None and pack.valid_for(params):
        return pack
    return PackedWeights(params)


_PARAM_ORDER = ['0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias', '6.weight', '6.bias', '8.weight', '8.bias',
                '10.weight', '10.bias', '12.weight', '12.bias', '14.weight', '14.bias']


# ---- the first half of the sweep's compact backward, ahead of time ---------------------------------------------------------------------------
# d3h_sdf_mlp_bwd_prepare (csrc/sdf_mlp_bwd.hip): once the caller knows which points CAN receive a gradient (hmsdf.tick_init: the vertices on
# sign-changing edges, marked by the regulariser's forward) the point list, the gather and the recompute of their activations do not have to
# wait for the gradient: they run on a stream of their own beside the loss / image-space backward instead of in the serial tail of the step
# (three launches, ~100 us at 9 k points).  D3H_SDF_PREPARE=0: everything inside the backward, as before.
PREPARE = os.environ.get('D3H_SDF_PREPARE', '1') != '0'
_LAST_SWEEP = [None]          # the state of the most recent differentiable sweep: (sdf storage pointer, state dict)
_PREP_STREAM = {}


def prepare_backward(sdf, edges32):
    """queue the first half of the compact backward of the sweep that produced `sdf` (see above) on the prepare stream: mark the vertices on
    sign-changing edges (the edge visit of the SDF regulariser, csrc/image_ops.hip:sdf_reg_fwd_kernel, run for its marks only), list them,
    gather them, recompute their activations.  A no-op when that sweep cannot use it."""
    st = _LAST_SWEEP[0]
    if not PREPARE or st is None or st[0] != sdf.data_ptr() or st[1].get('done') or edges32 is None:
        return
    state = st[1]
    lib = L.lib()
    n, dev = state['n'], sdf.device
    if sdf.numel() != n:
        return
    x, deform, disp, wp3_rec = state['x'], state['deform'], state['disp'], state['wp3_rec']
    main = _cur_stream() if (sdf.is_cuda and not L.emulated()) else None
    side = None
    if main is not None:
        side = _PREP_STREAM.get(dev)
        if side is None:
            side = _PREP_STREAM[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)
    import contextlib
    sd = sdf.detach().reshape(-1)
    e32 = edges32.contiguous()
    with (L.use_stream(side) if side is not None else contextlib.nullcontext()):
        marks = torch.zeros(n, dtype=torch.float32, device=dev)
        sums = torch.empty(2, dtype=torch.float32, device=dev)
        act = torch.empty(int(lib.d3h_sdf_mlp_act_floats(n)), dtype=torch.float32, device=dev)
        tiles = torch.empty(int(lib.d3h_sdf_mlp_bwd_scratch_ints(n)), dtype=torch.int32, device=dev)
        L.check(lib.d3h_sdf_reg_fwd(L.ptr(sd), L.ptr(e32), L.i32(e32.shape[0]), L.ptr(sums), L.ptr(marks), L.stream()), 'sdf_reg_fwd (marks)')
        L.check(lib.d3h_sdf_mlp_bwd_prepare(L.ptr(x), L.ptr(deform), L.f32(disp), L.ptr(marks), L.i64(n), L.ptr(tiles), L.ptr(wp3_rec),
                                            L.i32(_planes(wp3_rec)), L.ptr(act), L.stream()), 'sdf_mlp_bwd_prepare')
        ev = None
        if side is not None:
            ev = torch.cuda.Event()
            ev.record()
            for t in (x, deform, sd, e32, wp3_rec):
                if t is not None:
                    t.record_stream(side)
    state.update(done=True, act=act, tiles=tiles, event=ev)


class _SDFMLPFn(torch.autograd.Function):
    """sdf = MLP(x + disp*deform); first-order autograd only (the eikonal term's double backward uses the second-order ops below).
    `flat` is PackedWeights.flat (the arena-order parameter vector); the gradient w.r.t. it is the arena the kernels wrote."""

    @staticmethod
    def forward(ctx, x, deform, disp, pk, flat, rows):
        need = any(t is not None and t.requires_grad for t in (x, deform, flat))
        # rows = (lo, hi): evaluate x[lo:hi] only (a rank's shard of the grid sweep, d3h.dist_ops).  The FULL tensors are the inputs of
        # the node, so the gradient of `deform` is written into its full-size buffer directly -- no slice node with its zero-filled copy
        xs, ds = (x, deform) if rows is None else (x[rows[0]:rows[1]], deform[rows[0]:rows[1]] if deform is not None else None)
        if need:
            wp3 = _part(pk.wpf, 'fwd')
            ctx.wp3_rec = wp3 if (RECOMPUTE and wp3 is not None and _part(pk.wpt3, 'bwd') is not None) else None
            if ctx.wp3_rec is not None:
                sdf = forward(xs, pk.wp, deform=ds, disp=disp, wp3=wp3)
                act = xs.new_empty(0)
            else:
                sdf, act, _ = forward(xs, pk.wp, deform=ds, disp=disp, save=True, wp3=wp3)
            ctx.wpt, ctx.w14, ctx.wpt3 = pk.wpt, pk.w14, _part(pk.wpt3, 'bwd')
            ctx.save_for_backward(x, deform if deform is not None else x.new_empty(0), act)
            ctx.disp = float(disp)
            ctx.has_deform = deform is not None
            ctx.rows = rows
            ctx.deform_leaf = deform if (deform is not None and deform.is_leaf) else None
            ctx.prep = None
            if PREPARE and SPARSE_BACKWARD and ctx.wp3_rec is not None and rows is None:
                # (the whole-grid sweep of a single rank, without the activation save: its backward is the compact form and can be prepared)
                xc = x.contiguous().float()
                ctx.prep = {'n': int(xc.shape[0]), 'x': xc, 'deform': deform.contiguous().float() if deform is not None else None,
                            'disp': float(disp), 'wp3_rec': ctx.wp3_rec}
                _LAST_SWEEP[0] = (sdf.data_ptr(), ctx.prep)
        else:
            sdf = forward(xs, pk.wp, deform=ds, disp=disp, wp3=pk.wpf)
        return sdf.unsqueeze(-1)

    @staticmethod
    def backward(ctx, gout):
        x, deform, act = ctx.saved_tensors
        if not ctx.has_deform:
            deform = None
        lib = L.lib()
        wpt, w7, wpt3, wp3_rec = ctx.wpt, ctx.w14, ctx.wpt3, ctx.wp3_rec
        ctx.wpt = ctx.w14 = ctx.wpt3 = ctx.wp3_rec = None
        rows, leaf = ctx.rows, ctx.deform_leaf
        ctx.deform_leaf = None
        x_full, deform_full = x, deform
        if rows is not None:
            x, deform = x[rows[0]:rows[1]], (deform[rows[0]:rows[1]] if deform is not None else None)
        n = x.shape[0]
        dev = x.device
        xc = x.contiguous().float()
        g = gout.reshape(-1).contiguous().float()
        prep, ctx.prep = getattr(ctx, 'prep', None), None
        if _LAST_SWEEP[0] is not None and _LAST_SWEEP[0][1] is prep:
            _LAST_SWEEP[0] = None
        prepared = bool(prep and prep.get('done'))
        tiles = None
        if prepared:                        # list, gathered points and recomputed activations are in place (prepare_backward): join its stream
            act, tiles = prep['act'], prep['tiles']
            if prep['event'] is not None:
                _cur_stream().wait_event(prep['event'])
                for t in (act, tiles):
                    t.record_stream(_cur_stream())
        elif wp3_rec is not None:             # scratch for the recomputed activations (written at the visited tiles' own positions only)
            act = torch.empty(int(lib.d3h_sdf_mlp_act_floats(n)), dtype=torch.float32, device=dev)
        dz = torch.empty_like(act)
        dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
        arena = L.zeros(ARENA_FLOATS, torch.float32, dev)          # (carved from the zero slab: no fill launch); returned as d(flat)
        dw0, db0, dwh, dbh, dw4, db4, dw7, db7 = arena_views(arena)
        dfm = deform.contiguous().float() if deform is not None else None
        # active-tile list: the backward only visits 16-point tiles with a non-zero upstream gradient (csrc/sdf_mlp_bwd.hip, section 0)
        # scratch of the sparse backward: the position-tile list, or -- compact form, when the activations are recomputed -- the gathered problem
        if tiles is None:
            tiles = torch.empty(int(lib.d3h_sdf_mlp_bwd_scratch_ints(n)), dtype=torch.int32, device=dev) if SPARSE_BACKWARD else None
        L.check(lib.d3h_sdf_mlp_bwd(L.ptr(xc), L.ptr(dfm), L.f32(ctx.disp), L.ptr(g), L.ptr(w7), L.ptr(wpt), L.ptr(wpt3), L.ptr(act), L.ptr(dz),
                                    L.i64(n), L.ptr(dx), L.ptr(dw0), L.ptr(db0), L.ptr(dwh), L.ptr(dbh), L.ptr(dw4), L.ptr(db4),
                                    L.ptr(dw7), L.ptr(db7), L.ptr(tiles), L.ptr(wp3_rec), L.i32(_planes(wp3_rec)), L.i32(_planes(wpt3)),
                                    L.i32(1 if prepared else 0), L.stream()), 'sdf_mlp_bwd')
        d_deform = None
        if deform is not None and ctx.needs_input_grad[1]:
            # frame-parallel step: into the gradient's slice of the all-reduce arena (first contribution: written; later: added in place)
            d_deform = _GA.deliver(leaf if leaf is not None else deform_full, dx, ctx.disp, rows=rows)
        if rows is not None and ctx.needs_input_grad[0]:
            dxf = L.zeros_like(x_full, dtype=torch.float32)
            dxf[rows[0]:rows[1]] = dx
            dx = dxf
        return (dx if ctx.needs_input_grad[0] else None, d_deform, None, None, arena, None)


def sdf_query(x, params, deform=None, disp=0.0, pack=None, rows=None):
    """x[n,3] (+ disp*deform) -> sdf[n,1]; params: the 16 tensors of MLP.net in state_dict order; pack: a PackedWeights of them;
    rows = (lo, hi): only x[lo:hi] (-> sdf[hi-lo,1]), with the gradients landing in the full-size buffers"""
    pk = _packs(pack, params)
    return _SDFMLPFn.apply(x, deform, disp, pk, pk.flat, None if rows is None else (int(rows[0]), int(rows[1])))


class _SDFGradFn(torch.autograd.Function):
    """g = d(sdf)/d(x) at constant points x[n,3], differentiable w.r.t. the MLP parameters (the eikonal term of
    geometry/hmsdf.py:856-876: autograd.grad(..., create_graph=True) followed by a backward through the gradient graph).
    forward = fused forward with activation save + the first-order data backward with d(sdf) = 1; backward = the hand-derived
    second-order pass d3h_sdf_mlp_eik_bwd (tangent sweep, reverse sweep with the softplus'' injection, weight-gradient GEMMs)."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = L.lib()
        sd = {k: p for k, p in zip(_PARAM_ORDER, params)}
        wp = pack_weights(sd, prefix='')
        wpt = pack_weights_t(sd, prefix='')
        wp3 = pack_weights3(sd, prefix='') if X3 else None
        wpt3 = pack_weights_t3(sd, prefix='') if X3 else None
        xc = x.detach().contiguous().float()
        n = xc.shape[0]
        _, act, _ = forward(xc, wp, save=True, wp3=(pack_weights_h2(sd, prefix='') if (X3 and H2) else wp3))
        dz = torch.empty_like(act)
        g = torch.empty(n, 3, dtype=torch.float32, device=xc.device)
        w7 = sd['14.weight'].detach().contiguous().float()
        L.check(lib.d3h_sdf_mlp_grad_x(L.ptr(xc), L.ptr(w7), L.ptr(wpt), L.ptr(wpt3), L.i32(_planes(wpt3)), L.ptr(act), L.ptr(dz), L.i64(n), L.ptr(g), L.i32(0), L.stream()), 'sdf_mlp_grad_x')
        ctx.bufs = (xc, wp, wpt, wp3, wpt3, act, dz)
        return g

    @staticmethod
    def backward(ctx, u):
        lib = L.lib()
        xc, wp, wpt, wp3, wpt3, act, dz = ctx.bufs
        ctx.bufs = None
        n = xc.shape[0]
        dev = xc.device
        u = u.contiguous().float()
        tb, eb = torch.empty_like(act), torch.empty_like(act)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        dw0, db0, dwh, dbh, dw4, db4, dw7 = z(256, 39), z(256), z(5, 256, 256), z(5, 256), z(256, 295), z(256), z(1, 256)
        L.check(lib.d3h_sdf_mlp_eik_bwd(L.ptr(xc), L.ptr(u), L.ptr(wp), L.ptr(wpt), L.ptr(wp3), L.i32(_planes(wp3)), L.ptr(wpt3), L.i32(_planes(wpt3)), L.f32(0.0), L.ptr(act), L.ptr(dz), L.ptr(tb), L.ptr(eb), L.i64(n),
                                        L.ptr(dw0), L.ptr(db0), L.ptr(dwh), L.ptr(dbh), L.ptr(dw4), L.ptr(db4), L.ptr(dw7), L.i32(0), L.stream()),
                'sdf_mlp_eik_bwd')
        grads = [dw0, db0, dwh[0], dbh[0], dwh[1], dbh[1], dwh[2], dbh[2], dw4, db4, dwh[3], dbh[3], dwh[4], dbh[4], dw7, None]
        return (None, *grads)


def sdf_gradient(x, params):
    """x[n,3] (treated as constants) -> d(sdf)/d(x) [n,3], differentiable w.r.t. `params` (the 16 tensors of MLP.net)."""
    return _SDFGradFn.apply(x, *params)


LOSS_READY = None      # hand-over slot between _EikonalLossFn.forward and eikonal_loss() below, which attaches the event to ITS result


class _EikonalLossFn(torch.autograd.Function):
    """coeff * mean((|grad_x sdf(x)| - 1)^2) at constant points (hmsdf.py:856-876) with the parameter gradients computed EAGERLY in the
    forward: the term is linear in its upstream gradient, so backward only scales the stored gradients.  This moves the second-order
    sweeps (tangent, injected reverse, weight-gradient GEMMs: ~3 ms at 50 000 points) out of the GPU-saturated backward phase of the
    iteration into the forward phase, where they fill the host-bound gaps of the render / loss bookkeeping on the side stream."""

    @staticmethod
    def forward(ctx, x, coeff, pk, flat, begun, max_cus):
        lib = L.lib()
        wp, wpt = pk.wp, pk.wpt
        if begun is not None:               # eikonal_begin() already queued the forward sweep on these points with these weights
            xc, act = begun
        else:
            xc = x.detach().contiguous().float()
            _, act, _ = forward(xc, wp, save=True, max_cus=max_cus, wp3=_part(pk.wpf, 'eikfwd'))
        n = xc.shape[0]
        dev = xc.device
        dz = torch.empty_like(act)
        g = torch.empty(n, 3, dtype=torch.float32, device=dev)
        w7 = pk.w14
        wpt3 = _part(pk.wpt3, 'eik')
        L.check(lib.d3h_sdf_mlp_grad_x(L.ptr(xc), L.ptr(w7), L.ptr(wpt), L.ptr(wpt3), L.i32(_planes(wpt3)), L.ptr(act), L.ptr(dz), L.i64(n), L.ptr(g), L.i32(max_cus), L.stream()), 'sdf_mlp_grad_x')
        need = flat.requires_grad
        s = torch.empty(1, dtype=torch.float32, device=dev)
        u = torch.empty_like(g) if need else None
        L.check(lib.d3h_eikonal_loss(L.ptr(g), L.i64(n), L.f32(float(coeff) / max(n, 1)), L.ptr(s), L.ptr(u), L.stream()), 'eikonal_loss')
        ret = s[0] * (float(coeff) / max(n, 1))
        # The loss value exists now; the eager second-order sweeps below only produce parameter gradients.  A caller that runs this op
        # on a side stream can wait for LOSS_READY instead of the whole stream: the sweeps (MFMA-bound, ~2.3 ms at 5 10^4 points) then
        # keep running under the HBM-bound loss / render backward of the main stream.  The backward of this node is replayed on the
        # stream it was recorded on, i.e. after them.
        global LOSS_READY
        LOSS_READY = None
        if xc.is_cuda and not L.emulated():
            LOSS_READY = torch.cuda.Event()
            LOSS_READY.record()
            # inputs that were allocated on another stream and are read by the sweeps below: without this the caching allocator may
            # hand their memory out again as soon as the caller drops them (e.g. the next SDF sweep of the iteration re-packs the
            # weights) while this stream is still reading -- the caller no longer waits for the whole stream
            cur = _cur_stream()
            for t in (xc, wp, wpt, w7, pk.wp3, pk.wpt3, pk.wph):
                if t is None:
                    continue
                t.record_stream(cur)
        if need:
            tb, eb = torch.empty_like(act), torch.empty_like(act)
            # all parameter gradients live in ONE arena-order buffer (the output bias has none: its slot stays zero): backward scales it
            # with a single elementwise kernel and returns it as d(flat)
            arena = L.zeros(ARENA_FLOATS, torch.float32, dev)
            dw0, db0, dwh, dbh, dw4, db4, dw7, _ = arena_views(arena)
            wpj = _part(pk.wph if (pk.wph is not None and H2_JVP) else pk.wp3, 'eik')          # the tangent sweep's pack: fp16 x 2 when built
            L.check(lib.d3h_sdf_mlp_eik_bwd(L.ptr(xc), L.ptr(u), L.ptr(wp), L.ptr(wpt), L.ptr(wpj), L.i32(_planes(wpj)), L.ptr(wpt3), L.i32(_planes(wpt3)),
                                            L.f32(2.0 * float(coeff) / max(n, 1)), L.ptr(act), L.ptr(dz), L.ptr(tb), L.ptr(eb), L.i64(n),
                                            L.ptr(dw0), L.ptr(db0), L.ptr(dwh), L.ptr(dbh), L.ptr(dw4), L.ptr(db4), L.ptr(dw7), L.i32(max_cus), L.stream()),
                    'sdf_mlp_eik_bwd')
            ctx.arena = arena
        return ret

    @staticmethod
    def backward(ctx, gout):
        g = ctx.arena * gout
        ctx.arena = None
        return (None, None, None, g, None, None)


def eikonal_begin(x, params, pack=None, max_cus=0):
    """First kernel of eikonal_loss (the forward sweep with the activation save) on its own, so that a caller can queue it, issue other
    work while it runs (it is the longest single launch of the chain), and come back with eikonal_loss(..., begun=<this>)."""
    pk = _packs(pack, params)
    xc = x.detach().contiguous().float()
    _, act, _ = forward(xc, pk.wp, save=True, max_cus=max_cus, wp3=_part(pk.wpf, 'eikfwd'))
    return (xc, act)


def eikonal_loss(x, params, coeff, pack=None, begun=None, max_cus=0):
    """coeff * mean((|d sdf / d x| - 1)^2) over the points x[n,3] (constants); differentiable w.r.t. `params`.
    On the GPU the result carries `.d3h_ready`: an event recorded when the loss VALUE is complete (the eager second-order sweeps that
    follow it on the same stream only produce parameter gradients) -- one event per call, so two launches in flight cannot be confused."""
    global LOSS_READY
    LOSS_READY = None
    pk = _packs(pack, params)
    out = _EikonalLossFn.apply(x, float(coeff), pk, pk.flat, begun, int(max_cus))
    out.d3h_ready, LOSS_READY = LOSS_READY, None
    return out
