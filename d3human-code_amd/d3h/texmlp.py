"""Host side of csrc/texmlp.hip: multiresolution grid encoding (tcnn HashGrid, dense levels) + the kd/ks texture MLP."""
import ctypes
import os
import math
import weakref

import torch
from d3h._lib import cur_stream as _cur_stream

from . import _lib as L
from . import gradarena as _GA

PER_LEVEL_SCALE = math.exp(math.log(4096 / 16) / (16 - 1))     # render/mlptexture.py:62-65 -> 1.4472692374403782
BASE_RES = 16
N_LEVELS, N_FEATURES = 5, 2
ENC_DIMS = N_LEVELS * N_FEATURES


def grid_param_count(per_level_scale=PER_LEVEL_SCALE, base_res=BASE_RES):
    return int(L.lib().d3h_hashgrid_param_floats(ctypes.c_double(per_level_scale), L.i32(base_res)))


def _f6(v):
    return (ctypes.c_float * 6)(*[float(x) for x in v])


ASYNC_TABLE_GRAD = os.environ.get('D3H_ASYNC_TABLE_GRAD', '1') != '0'      # table-gradient scatter of the backward on its own stream (see _TexMLPFn.backward)
_SIDE = {}
_PENDING = []
# One backward pass may hold several nodes that contribute to the SAME table (shade() samples twice for kd_grad / ks_grad, the split
# stage renders twice).  The engine sums their table gradients on the main stream the moment the second one is returned, and it
# knows nothing about the scatter stream: only the FIRST table gradient of a pass may still be in flight when it is handed over
# (every later node joins the scatter stream before it returns anything, so the sum reads finished data), and only while the leaf
# has no .grad yet (AccumulateGrad then keeps the tensor instead of adding to it on the main stream).
_PASS = {'table_grad_returned': False, 'callback': False}


def _scatter_stream(t):
    if not t.is_cuda or L.emulated():
        return None
    s = _SIDE.get(t.device)
    if s is None:
        s = _SIDE[t.device] = torch.cuda.Stream(device=t.device)
    return s


def _join_scatter():
    while _PENDING:
        _cur_stream().wait_stream(_PENDING.pop())


def _end_of_pass():
    _join_scatter()
    _PASS['table_grad_returned'] = False
    _PASS['callback'] = False


def _mark_table_grad_returned():
    """called by every backward node that returns a table gradient; arms the end-of-pass reset once per backward pass"""
    _PASS['table_grad_returned'] = True
    if not _PASS['callback']:
        _PASS['callback'] = True
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass)


class _TexMLPFn(torch.autograd.Function):
    """out[n,6] = sigmoid(MLP(grid_encode(clamp((x - b0)/(b1 - b0), 0, 1)))) * (omax - omin) + omin"""

    @staticmethod
    def forward(ctx, x, mask, table, w1, w2, w3, bbox, omin, omax, in_grad_scale):
        xs = x.reshape(-1, 3).contiguous().float()
        n = xs.shape[0]
        wcat = torch.cat([w1.reshape(-1), w2.reshape(-1), w3.reshape(-1)]).contiguous().float()
        tab = table.contiguous().float()
        m = mask.reshape(-1).contiguous().float() if mask is not None else None
        out = torch.empty(n, 6, dtype=torch.float32, device=x.device)
        L.check(L.lib().d3h_texmlp_fwd(L.ptr(xs), L.ptr(m), L.ptr(tab), L.ptr(wcat), L.i64(n), ctypes.c_double(PER_LEVEL_SCALE), L.i32(BASE_RES),
                                       _f6(bbox), _f6(omin), _f6(omax), L.ptr(out), None, L.stream()), 'texmlp_fwd')
        ctx.save_for_backward(xs, m if m is not None else xs.new_empty(0), tab, wcat)
        ctx.table_leaf = weakref.ref(table) if table.is_leaf else None
        ctx.w_leaves = (w1, w2, w3) if all(w.is_leaf and w.dtype == torch.float32 and w.is_contiguous() for w in (w1, w2, w3)) else None
        ctx.meta = (bbox, omin, omax, float(in_grad_scale), mask is not None, x.shape, w1.shape, w2.shape, w3.shape)
        return out.reshape(*x.shape[:-1], 6)

    @staticmethod
    def backward(ctx, g):
        xs, m, tab, wcat = ctx.saved_tensors
        bbox, omin, omax, gs, has_mask, xshape, s1, s2, s3 = ctx.meta
        n = xs.shape[0]
        lib = L.lib()
        need_tab, need_w, need_x = ctx.needs_input_grad[2], any(ctx.needs_input_grad[3:6]), ctx.needs_input_grad[0]
        # Frame-parallel step: leaf gradients are produced inside the step's all-reduce arena (d3h.gradarena).  The first contribution of a
        # pass takes the parameter's (pre-zeroed) slice and returns it; a later one accumulates onto the slice with the same atomics and
        # returns nothing, so autograd has no 4.3 MB sum to form.
        leaf = ctx.table_leaf() if ctx.table_leaf is not None else None
        wl, ctx.w_leaves = ctx.w_leaves, None
        d_tab = d_tab_ret = d_w = None
        w_ret = True
        first_contribution = not _PASS['table_grad_returned'] and leaf is not None and leaf.grad is None
        if need_tab:
            d_tab = d_tab_ret = _GA.slot_for(leaf) if tab.data_ptr() == (leaf.data_ptr() if leaf is not None else 0) else None
            if d_tab is None:
                d_tab = _GA.accum_for(leaf)
                if d_tab is None:
                    d_tab = d_tab_ret = L.zeros_like(tab)
        if need_w:
            if wl is not None and all(ctx.needs_input_grad[3:6]):
                d_w = _GA.block_for(wl)
                if d_w is None:
                    d_w = _GA.accum_block_for(wl)
                    w_ret = d_w is None
            if d_w is None:
                d_w = L.zeros_like(wcat)
        d_x = torch.empty_like(xs) if need_x else None
        genc = torch.empty(n, 10, dtype=torch.float32, device=xs.device)       # d(encoding) between the two halves of the split backward
        gc = g.reshape(-1, 6).contiguous().float()
        mp = m if has_mask else None
        args = lambda: (L.ptr(xs), L.ptr(mp), L.ptr(tab), L.ptr(wcat), L.i64(n), ctypes.c_double(PER_LEVEL_SCALE), L.i32(BASE_RES), _f6(bbox),
                        _f6(omin), _f6(omax), L.f32(gs))
        side = _scatter_stream(xs) if (ASYNC_TABLE_GRAD and need_tab and need_x and first_contribution) else None
        if side is None:
            _join_scatter()            # an earlier node's table gradient may still be in flight: the engine is about to add ours to it
            L.check(lib.d3h_texmlp_bwd(*args(), L.i32(0), L.ptr(gc), L.ptr(d_tab), L.ptr(d_w), L.ptr(d_x), L.ptr(genc), L.stream()), 'texmlp_bwd')
        else:
            # The table gradient is a leaf result and its scatter is bound by fabric atomics (0.6 ms at 4 x 1024^2), while the position
            # gradient is on the critical path of the backward.  MLP half and position gradient on this stream; the scatter on a second
            # stream, where it co-runs with the rest of the backward (rasteriser, LBS, the MFMA-bound SDF backward).  The pass joins
            # the stream when it ends (queue_callback), a second contribution to the same table joins it first.
            main = _cur_stream()
            _join_scatter()
            L.check(lib.d3h_texmlp_bwd(*args(), L.i32(0), L.ptr(gc), None, L.ptr(d_w), None, L.ptr(genc), L.stream()), 'texmlp_bwd_mlp')
            L.check(lib.d3h_texmlp_bwd(*args(), L.i32(1), L.ptr(genc), None, None, L.ptr(d_x), None, L.stream()), 'texmlp_bwd_dx')
            side.wait_stream(main)
            with L.use_stream(side):
                L.check(lib.d3h_texmlp_bwd(*args(), L.i32(1), L.ptr(genc), L.ptr(d_tab), None, None, None, L.stream()), 'texmlp_bwd_table')
            for t in (xs, tab, genc, d_tab) + ((mp,) if mp is not None else ()):
                t.record_stream(side)
            _PENDING.append(side)
        if need_tab:
            _mark_table_grad_returned()
        if d_w is not None and w_ret:
            n1, n2 = s1.numel(), s2.numel()
            dw1, dw2, dw3 = d_w[:n1].reshape(s1), d_w[n1:n1 + n2].reshape(s2), d_w[n1 + n2:].reshape(s3)
        else:
            dw1 = dw2 = dw3 = None
        return (d_x.reshape(xshape) if d_x is not None else None, None, d_tab_ret, dw1, dw2, dw3, None, None, None, None)


def texture_mlp(x, table, w1, w2, w3, bbox, omin, omax, mask=None, in_grad_scale=128.0):
    """x [...,3] world positions -> [...,6]; mask [...] (optional): pixels with mask <= 0 are skipped (zeros, no gradient)"""
    return _TexMLPFn.apply(x, mask, table, w1, w2, w3, tuple(float(v) for v in bbox), tuple(float(v) for v in omin),
                           tuple(float(v) for v in omax), in_grad_scale)


class _GridEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table):
        xs = x.reshape(-1, 3).contiguous().float()
        n = xs.shape[0]
        tab = table.contiguous().float()
        enc = torch.empty(n, ENC_DIMS, dtype=torch.float32, device=x.device)
        unit = (0.0, 0.0, 0.0, 1.0, 1.0, 1.0)
        L.check(L.lib().d3h_texmlp_fwd(L.ptr(xs), None, L.ptr(tab), None, L.i64(n), ctypes.c_double(PER_LEVEL_SCALE), L.i32(BASE_RES), _f6(unit),
                                       None, None, None, L.ptr(enc), L.stream()), 'hashgrid_fwd')
        ctx.save_for_backward(xs, tab)
        ctx.xshape = x.shape
        return enc

    @staticmethod
    def backward(ctx, g):
        xs, tab = ctx.saved_tensors
        _join_scatter()
        if ctx.needs_input_grad[1]:
            _mark_table_grad_returned()
        d_tab = L.zeros_like(tab) if ctx.needs_input_grad[1] else None
        d_x = torch.empty_like(xs) if ctx.needs_input_grad[0] else None
        unit = (0.0, 0.0, 0.0, 1.0, 1.0, 1.0)
        L.check(L.lib().d3h_texmlp_bwd(L.ptr(xs), None, L.ptr(tab), None, L.i64(xs.shape[0]), ctypes.c_double(PER_LEVEL_SCALE), L.i32(BASE_RES),
                                       _f6(unit), None, None, L.f32(1.0), L.i32(1), L.ptr(g.contiguous().float()), L.ptr(d_tab), None, L.ptr(d_x),
                                       None, L.stream()), 'hashgrid_bwd')
        return (d_x.reshape(ctx.xshape) if d_x is not None else None), d_tab


def grid_encode(x, table):
    """tcnn.Encoding(3, HashGrid).forward for x in [0,1]^3 -> [n,10] (fp32)"""
    return _GridEncodeFn.apply(x, table)
