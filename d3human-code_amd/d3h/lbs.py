"""Host side of csrc/lbs.hip: nearest-template-vertex skin weights + blended inverse/forward skinning.

Mirrors deform/smplx_exavatar_deformer.py: interpolate_weights :363-383 (K=1), apply_lbs_inverse :385-421,
lbs_forward :434-486 -- for a whole batch of frames at once (the nearest-vertex ids depend only on the canonical
points, so they are computed once and shared by every frame).
"""
import torch

from . import _lib as L


def knn1(pts, tmpl):
    """index (int32 [P]) of the nearest template vertex; squared L2, first minimum wins (knn_cpu.cpp:13-69)"""
    lib = L.lib()
    pts = pts.detach().contiguous().float()
    tmpl = tmpl.detach().contiguous().float()
    idx = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    L.check(lib.d3h_knn1(L.ptr(pts), L.i32(pts.shape[0]), L.ptr(tmpl), L.i32(tmpl.shape[0]), L.ptr(idx), None, L.stream()), 'knn1')
    return idx


class _LBSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, idx, lbs_w, A0, A, trans):
        lib = L.lib()
        pts_c = pts.contiguous().float()
        A0c = A0.detach().reshape(-1, 16).contiguous().float()
        Ac = A.detach().reshape(A.shape[0], -1, 16).contiguous().float()
        tr = trans.detach().reshape(-1, 3).contiguous().float()
        nb, nj, P = Ac.shape[0], Ac.shape[1], pts_c.shape[0]
        out = torch.empty(nb, P, 3, dtype=torch.float32, device=pts.device)
        L.check(lib.d3h_lbs_fwd(L.ptr(pts_c), L.i32(P), L.ptr(idx), L.ptr(lbs_w), L.i32(nj), L.ptr(A0c), L.ptr(Ac), L.ptr(tr),
                                L.i32(nb), L.ptr(out), None, L.stream()), 'lbs_fwd')
        ctx.save_for_backward(pts_c, idx, lbs_w, A0c, Ac)
        ctx.shapes = (A.shape, trans.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        pts, idx, lbs_w, A0c, Ac = ctx.saved_tensors
        lib = L.lib()
        nb, nj, P = Ac.shape[0], Ac.shape[1], pts.shape[0]
        g = g.contiguous().float()
        d_pts = torch.zeros_like(pts)
        need_A, need_t = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        dA = torch.zeros(nb, nj, 16, dtype=torch.float32, device=pts.device) if need_A else None
        dT = torch.zeros(nb, 3, dtype=torch.float32, device=pts.device) if need_t else None
        L.check(lib.d3h_lbs_bwd(L.ptr(pts), L.i32(P), L.ptr(idx), L.ptr(lbs_w), L.i32(nj), L.ptr(A0c), L.ptr(Ac), L.i32(nb), L.ptr(g),
                                L.ptr(d_pts), L.ptr(dA), L.ptr(dT), L.stream()), 'lbs_bwd')
        a_shape, t_shape = ctx.shapes
        return (d_pts, None, None, None, dA.reshape(a_shape) if need_A else None, dT.reshape(t_shape) if need_t else None)


def lbs_points(pts, idx, lbs_w, A0, A, trans):
    """pts [P,3] canonical-mesh points, idx [P] nearest template vertex, lbs_w [V,J], A0 [J,4,4] init-pose transforms,
    A [B,J,4,4] frame transforms, trans [B,3]  ->  posed points [B,P,3]"""
    return _LBSFn.apply(pts, idx, lbs_w.contiguous().float(), A0, A, trans)
