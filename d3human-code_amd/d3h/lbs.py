"""Host side of csrc/lbs.hip: nearest-template-vertex skin weights + blended inverse/forward skinning.

Mirrors deform/smplx_exavatar_deformer.py: interpolate_weights :363-383 (K=1), apply_lbs_inverse :385-421,
lbs_forward :434-486 -- for a whole batch of frames at once (the nearest-vertex ids depend only on the canonical
points, so they are computed once and shared by every frame).
"""
import ctypes

import torch

from . import _lib as L


class KnnGrid:
    """uniform-grid binning of a fixed template for d3h_knn1_grid (csrc/lbs.hip): built once (a sort of the template), reused by every
    iteration.  The query returns exactly what the exhaustive d3h_knn1 returns."""

    def __init__(self, tmpl, max_cells_per_axis=128):
        t = tmpl.detach().contiguous().float()
        nv = t.shape[0]
        lo, hi = t.min(0).values, t.max(0).values
        ext = (hi - lo).clamp_min(1e-6)
        h = float((ext.prod() / max(nv, 1)) ** (1.0 / 3.0))
        h = max(h, float(ext.max()) / max_cells_per_axis) * 1.0001
        lo = lo - 0.5 * h                                               # every vertex strictly inside the box
        g = [int(v) for v in torch.ceil((hi - lo) / h + 0.5).clamp(1, max_cells_per_axis + 2).tolist()]
        self.h, self.g = h, g
        self.lo = (ctypes.c_float * 3)(*[float(v) for v in lo.tolist()])     # HOST argument of the C ABI
        lo = torch.tensor(list(self.lo), dtype=torch.float32, device=t.device)
        inv_h = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(h, dtype=torch.float32)
        c = torch.floor((t - lo) * inv_h.to(t.device)).long()
        for a in range(3):
            c[:, a].clamp_(0, g[a] - 1)
        cell = (c[:, 2] * g[1] + c[:, 1]) * g[0] + c[:, 0]
        order = torch.sort(cell, stable=True).indices                   # ascending original index inside a cell
        ncell = g[0] * g[1] * g[2]
        counts = torch.bincount(cell, minlength=ncell)
        self.cell_start = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).to(torch.int32).contiguous()
        self.cell_pts = torch.cat([t[order], order.to(torch.int32).view(torch.float32)[:, None]], dim=1).contiguous()
        # seed of every cell: sorted position of the vertex nearest to the cell centre (one exhaustive search, once)
        ii = [torch.arange(n, device=t.device, dtype=torch.float32) for n in g]
        cz, cy, cx = torch.meshgrid(ii[2], ii[1], ii[0], indexing='ij')
        centres = (torch.stack([cx, cy, cz], -1).reshape(-1, 3) + 0.5) * h + lo
        inv = torch.empty(nv, dtype=torch.int64, device=t.device)
        inv[order] = torch.arange(nv, device=t.device)
        self.cell_seed = inv[knn1(centres.contiguous(), t).long()].to(torch.int32).contiguous()
        self.nv = nv
        self.key = (tmpl.data_ptr(), tmpl._version, tuple(tmpl.shape))
        self.src = tmpl                      # kept alive: the address in the key cannot be handed to another tensor meanwhile

    def matches(self, tmpl):
        return self.key == (tmpl.data_ptr(), tmpl._version, tuple(tmpl.shape))

    def query_counted(self, pts_cap, counts):
        """nearest template vertex of the first counts[0] + 3 counts[1] + 4 counts[2] rows of `pts_cap` (a buffer at its capacity), the row
        count read on the DEVICE (d3h/mtets.py: work queued before the host knows the sizes of the extraction) -> idx [capacity] int32"""
        lib = L.lib()
        pts = pts_cap.detach()
        idx = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
        L.check(lib.d3h_knn1_grid_counted(L.ptr(pts), L.i32(pts.shape[0]), L.ptr(counts), L.ptr(self.cell_pts), L.ptr(self.cell_start),
                                          L.ptr(self.cell_seed), L.i32(self.nv), self.lo, L.f32(self.h), L.i32(self.g[0]), L.i32(self.g[1]),
                                          L.i32(self.g[2]), L.ptr(idx), L.stream()), 'knn1_grid_counted')
        return idx

    def query(self, pts, want_dist=False):
        lib = L.lib()
        pts = pts.detach().contiguous().float()
        idx = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
        dist = torch.empty(pts.shape[0], dtype=torch.float32, device=pts.device) if want_dist else None
        L.check(lib.d3h_knn1_grid(L.ptr(pts), L.i32(pts.shape[0]), L.ptr(self.cell_pts), L.ptr(self.cell_start), L.ptr(self.cell_seed), L.i32(self.nv),
                                  self.lo, L.f32(self.h), L.i32(self.g[0]), L.i32(self.g[1]), L.i32(self.g[2]), L.ptr(idx),
                                  L.ptr(dist), L.stream()), 'knn1_grid')
        return (idx, dist) if want_dist else idx


def knn1(pts, tmpl, grid=None):
    """index (int32 [P]) of the nearest template vertex; squared L2, first minimum wins (knn_cpu.cpp:13-69).  `grid`: a KnnGrid of
    `tmpl` (fixed templates: same result, ~10x faster)"""
    if grid is not None:
        return grid.query(pts)
    lib = L.lib()
    pts = pts.detach().contiguous().float()
    tmpl = tmpl.detach().contiguous().float()
    idx = torch.empty(pts.shape[0], dtype=torch.int32, device=pts.device)
    L.check(lib.d3h_knn1(L.ptr(pts), L.i32(pts.shape[0]), L.ptr(tmpl), L.i32(tmpl.shape[0]), L.ptr(idx), None, L.stream()), 'knn1')
    return idx


class _LBSFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts, idx, lbs_w, A0, A, trans, pre):
        lib = L.lib()
        pts_c = pts.contiguous().float()
        A0c = A0.detach().reshape(-1, 16).contiguous().float()
        Ac = A.detach().reshape(A.shape[0], -1, 16).contiguous().float()
        tr = trans.detach().reshape(-1, 3).contiguous().float()
        nb, nj, P = Ac.shape[0], Ac.shape[1], pts_c.shape[0]
        if pre is not None:                 # computed by lbs_points_counted from the same arguments before the row count was known on the host
            if tuple(pre.shape) != (nb, P, 3) or not pre.is_contiguous():
                raise RuntimeError(f'd3h lbs_points: the pre-computed result is {tuple(pre.shape)}, expected {(nb, P, 3)}')
            out = pre
        else:
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
        d_pts = torch.empty_like(pts)
        per_frame = torch.empty(nb, P, 3, dtype=torch.float32, device=pts.device) if nb > 1 else None      # summed in frame order: deterministic
        need_A, need_t = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
        dA = L.zeros((nb, nj, 16), torch.float32, pts.device) if need_A else None
        dT = L.zeros((nb, 3), torch.float32, pts.device) if need_t else None
        L.check(lib.d3h_lbs_bwd(L.ptr(pts), L.i32(P), L.ptr(idx), L.ptr(lbs_w), L.i32(nj), L.ptr(A0c), L.ptr(Ac), L.i32(nb), L.ptr(g),
                                L.ptr(d_pts), L.ptr(per_frame), L.ptr(dA), L.ptr(dT), L.stream()), 'lbs_bwd')
        a_shape, t_shape = ctx.shapes
        return (d_pts, None, None, None, dA.reshape(a_shape) if need_A else None, dT.reshape(t_shape) if need_t else None, None)


def lbs_points(pts, idx, lbs_w, A0, A, trans, pre=None):
    """pts [P,3] canonical-mesh points, idx [P] nearest template vertex, lbs_w [V,J], A0 [J,4,4] init-pose transforms,
    A [B,J,4,4] frame transforms, trans [B,3]  ->  posed points [B,P,3].  `pre`: the result already computed by lbs_points_counted (the
    launch is skipped, the autograd node is the same)"""
    return _LBSFn.apply(pts, idx, lbs_w.contiguous().float(), A0, A, trans, pre)


def lbs_points_counted(pts_cap, counts, idx_cap, lbs_w, A0, A, trans):
    """lbs_points over the first r = counts[0] + 3 counts[1] + 4 counts[2] rows of `pts_cap` [capacity, 3], r read on the DEVICE: the launch
    is queued before the host knows r.  -> a flat float buffer whose leading B * r * 3 floats are the dense [B, r, 3] result (narrow it with
    counted_result once r is known and hand it to lbs_points(..., pre=))"""
    lib = L.lib()
    A0c = A0.detach().reshape(-1, 16).contiguous().float()
    Ac = A.detach().reshape(A.shape[0], -1, 16).contiguous().float()
    tr = trans.detach().reshape(-1, 3).contiguous().float()
    nb, nj, cap = Ac.shape[0], Ac.shape[1], pts_cap.shape[0]
    out = torch.empty(nb * cap * 3, dtype=torch.float32, device=pts_cap.device)
    L.check(lib.d3h_lbs_fwd_counted(L.ptr(pts_cap.detach()), L.i32(cap), L.ptr(counts), L.ptr(idx_cap), L.ptr(lbs_w.contiguous().float()), L.i32(nj),
                                    L.ptr(A0c), L.ptr(Ac), L.ptr(tr), L.i32(nb), L.ptr(out), L.stream()), 'lbs_fwd_counted')
    return out


def counted_result(flat, nb, rows):
    return flat[:nb * rows * 3].view(nb, rows, 3)
