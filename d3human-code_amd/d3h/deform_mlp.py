"""Host side of the fused offset-network kernels (csrc/deform_mlp.hip, deform_mlp_bwd.hip): MLP_deform of the seq stage
(geometry/mlp.py:77-118; hmsdf.py:658-665) for its working shape n_freq 8, d_hidden 256, n_hidden 6, skip_in [3], d_out 3.

The 136-float pose code is constant over the points: W0 [code; emb] + b0 = W0[:, 136:] emb + (b0 + W0[:, :136] code).  The wrapper folds
it into the first bias before packing and unfolds the gradient afterwards:  d(b0) = d(b0'),  d(W0[:, :136]) = d(b0') (x) code,
d(code) = W0[:, :136]^T d(b0').  The points are treated as constants (they are the fixed base-mesh vertices in getMesh_seq)."""
import torch

from . import _lib as L

CODE = 136
EMB = 51
HIDDEN_KEYS = (2, 4, 6, 10, 12)


def _lib():
    return L.lib()


def supported(net):
    """True when `net` (geometry.mlp.MLP_deform) has the shape the kernels are built for"""
    want = {0: (256, CODE + EMB), 2: (256, 256), 4: (256, 256), 6: (256, 256), 8: (256, 256 + EMB), 10: (256, 256), 12: (256, 256), 14: (3, 256)}
    try:
        return len(net.net) == 15 and all(tuple(net.net[i].weight.shape) == s for i, s in want.items()) and not net.use_float16
    except (AttributeError, IndexError):
        return False


class _DeformMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, code, *params):
        lib = _lib()
        w = {i: params[2 * k] for k, i in enumerate((0, 2, 4, 6, 8, 10, 12, 14))}
        b = {i: params[2 * k + 1] for k, i in enumerate((0, 2, 4, 6, 8, 10, 12, 14))}
        c = code.reshape(CODE).float()
        W0 = w[0].detach().float()
        w0e = W0[:, CODE:].contiguous()
        b0f = (b[0].detach().float() + W0[:, :CODE] @ c.detach()).contiguous()
        wh = torch.stack([w[i].detach().float() for i in HIDDEN_KEYS]).contiguous()
        bh = torch.stack([b[i].detach().float() for i in HIDDEN_KEYS]).contiguous()
        w4, b4 = w[8].detach().contiguous().float(), b[8].detach().contiguous().float()
        w7, b7 = w[14].detach().contiguous().float(), b[14].detach().contiguous().float()
        xc = x.detach().reshape(-1, 3).contiguous().float()
        n, dev = xc.shape[0], xc.device
        wp = torch.empty(lib.d3h_deform_mlp_wpack_floats(), dtype=torch.float32, device=dev)
        L.check(lib.d3h_deform_mlp_pack(L.ptr(w0e), L.ptr(b0f), L.ptr(wh), L.ptr(bh), L.ptr(w4), L.ptr(b4), L.ptr(w7), L.ptr(b7), L.ptr(wp), L.stream()),
                'deform_mlp_pack')
        need = code.requires_grad or any(p.requires_grad for p in params)
        out = torch.empty(n, 3, dtype=torch.float32, device=dev)
        act = torch.empty(lib.d3h_deform_mlp_act_floats(n), dtype=torch.float32, device=dev) if need else None
        L.check(lib.d3h_deform_mlp_fwd(L.ptr(xc), None, L.f32(0.0), L.ptr(wp), L.ptr(out), None, L.ptr(act), L.i64(n), L.i32(0), L.stream()), 'deform_mlp_fwd')
        if need:
            ctx.save_for_backward(xc, c.detach(), W0, w0e, wh, w4, w7, act)
        ctx.xshape = x.shape
        return out.reshape(*x.shape[:-1], 3)

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        xc, c, W0, w0e, wh, w4, w7, act = ctx.saved_tensors
        n, dev = xc.shape[0], xc.device
        wpt = torch.empty(lib.d3h_deform_mlp_wpackt_floats(), dtype=torch.float32, device=dev)
        L.check(lib.d3h_deform_mlp_pack_t(L.ptr(w0e), L.ptr(wh), L.ptr(w4), L.ptr(wpt), L.stream()), 'deform_mlp_pack_t')
        gout = g.reshape(-1, 3).contiguous().float()
        dz = torch.empty_like(act)
        sizes = [256 * EMB, 256, 5 * 65536, 5 * 256, 256 * (256 + EMB), 256, 3 * 256, 3]
        dw0e, db0, dwh, dbh, dw4, db4, dw7, db7 = torch.split(torch.zeros(sum(sizes), dtype=torch.float32, device=dev), sizes)
        L.check(lib.d3h_deform_mlp_bwd(L.ptr(xc), None, L.f32(0.0), L.ptr(gout), L.ptr(w7), L.ptr(wpt), None, L.ptr(act), L.ptr(dz), L.i64(n), None,
                                       L.ptr(dw0e), L.ptr(db0), L.ptr(dwh), L.ptr(dbh), L.ptr(dw4), L.ptr(db4), L.ptr(dw7), L.ptr(db7), None,
                                       None, L.i32(0), L.i32(0), L.i32(0), L.stream()), 'deform_mlp_bwd')
        dW0 = torch.cat([torch.outer(db0, c), dw0e.view(256, EMB)], dim=1)            # unfold the pose code from the first bias
        dcode = (W0[:, :CODE].t() @ db0).reshape(1, 1, CODE)
        dwh, dbh = dwh.view(5, 256, 256), dbh.view(5, 256)
        grads = [dW0, db0, dwh[0], dbh[0], dwh[1], dbh[1], dwh[2], dbh[2], dw4.view(256, 256 + EMB), db4, dwh[3], dbh[3], dwh[4], dbh[4],
                 dw7.view(3, 256), db7]
        return (None, dcode, *grads)


def offsets(x, code, params):
    """x [..., 3] (constants), code [1,1,136], params: the 16 tensors of MLP_deform.net (weight, bias per Linear) -> offsets [..., 3]"""
    return _DeformMLPFn.apply(x, code, *params)
