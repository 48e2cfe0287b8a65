"""Host side of csrc/mesh_ops.hip: the seq-stage geometry regularisers on a fixed-topology mesh.

    laplacian_loss(v, topo)                  render/mesh.py:30-82 + lap_loss.py:40-47   mean |L V|^2, uniform Laplacian
    normal_consistency(v, faces32, pairs32)  render/mesh.py:18-28,266-279               mean (1 - cos)^2 over connected faces
    collision_loss(cloth, body, faces, eps)  geometry/hmsdf.py:98-132                   mean relu(eps - (p - c_f) . n_f)^2
    find_edges / find_connected_faces        render/mesh.py:85-134                      vectorised, on the tensors' device

`EdgeTopology` holds what the reference rebuilds from the edge list on every call (degree, adjacency): built once per edge tensor."""
import torch

from . import _lib as L
from . import lbs as _lbs


def find_edges(indices, remove_duplicates=True):
    e = torch.cat([indices[:, [0, 1]], indices[:, [1, 2]], indices[:, [2, 0]]], dim=1).view(indices.shape[0] * 3, 2)
    if remove_duplicates:
        e = torch.unique(torch.sort(e, dim=1).values, dim=0)
    return e


def find_connected_faces(indices):
    """mesh.py:106-134 without the per-edge Python loop: (pairs [E,2] of faces sharing an edge, first-seen face first, rows in ascending
    edge order; all 3F sorted edges).  Raises like the reference's assert when an edge has more than two faces."""
    e = torch.sort(find_edges(indices, remove_duplicates=False), dim=1).values
    _, inv, counts = torch.unique(e, dim=0, return_inverse=True, return_counts=True)
    if int(counts.max()) != 2:
        raise AssertionError('find_connected_faces: non-manifold or open edge (mesh.py:116 asserts counts.max() == 2)')
    face_ids = torch.arange(indices.shape[0], device=indices.device).repeat_interleave(3)
    order = torch.argsort(inv, stable=True)
    starts = torch.cumsum(counts, 0) - counts
    two = counts == 2
    pairs = torch.stack([face_ids[order[starts[two]]], face_ids[order[starts[two] + 1]]], dim=1)
    return pairs, e


class EdgeTopology:
    """CSR adjacency + 1/degree of a unique undirected edge list [E,2] over nv vertices (compute_laplacian_uniform's A and deg)"""
    _cache = {}

    def __init__(self, edges, nv):
        dev = edges.device
        e = edges.long()
        src = torch.cat([e[:, 0], e[:, 1]])
        dst = torch.cat([e[:, 1], e[:, 0]])
        order = torch.argsort(src, stable=True)
        deg = torch.bincount(src, minlength=nv)
        self.nv = nv
        self.offs = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(deg, 0)]).int().contiguous()
        self.nbr = dst[order].int().contiguous()
        degf = deg.float()
        self.inv_deg = torch.where(degf > 0, 1.0 / degf, degf).contiguous()

    @classmethod
    def get(cls, edges, nv):
        k = (edges.data_ptr(), tuple(edges.shape), nv, str(edges.device))
        t = cls._cache.get(k)
        if t is None or t[0] is not edges:
            if len(cls._cache) > 16:
                cls._cache.clear()
            t = (edges, cls(edges, nv))
            cls._cache[k] = t
        return t[1]


class _LaplacianLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, topo):
        vc = v.contiguous().float()
        nv = vc.shape[0]
        lv = torch.empty_like(vc)
        s = torch.empty(1, dtype=torch.float32, device=vc.device)
        L.check(L.lib().d3h_laplacian_loss_fwd(L.ptr(vc), L.i32(nv), L.ptr(topo.offs), L.ptr(topo.nbr), L.ptr(topo.inv_deg), L.ptr(lv), L.ptr(s),
                                               L.stream()), 'laplacian_loss_fwd')
        ctx.save_for_backward(lv)
        ctx.topo = topo
        return s[0] / nv

    @staticmethod
    def backward(ctx, g):
        lv, = ctx.saved_tensors
        topo = ctx.topo
        nv = lv.shape[0]
        d_v = torch.empty_like(lv)
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_laplacian_loss_bwd(L.ptr(lv), L.i32(nv), L.ptr(topo.offs), L.ptr(topo.nbr), L.ptr(topo.inv_deg), L.ptr(gs),
                                               L.f32(2.0 / nv), L.ptr(d_v), L.stream()), 'laplacian_loss_bwd')
        return d_v, None


def laplacian_loss(v, edges):
    """lap_loss.py:40-47 body_laplacian_loss: mean over vertices of |(L V)_i|^2 with mesh.py:30-82's uniform Laplacian"""
    return _LaplacianLossFn.apply(v, EdgeTopology.get(edges, v.shape[0]))


class _NormalConsistencyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, faces32, pairs32):
        vc = v.contiguous().float()
        n = pairs32.shape[0]
        s = torch.empty(1, dtype=torch.float32, device=vc.device)
        L.check(L.lib().d3h_normal_consistency_fwd(L.ptr(vc), L.ptr(faces32), L.ptr(pairs32), L.i32(n), L.ptr(s), L.stream()), 'normal_consistency_fwd')
        ctx.save_for_backward(vc, faces32, pairs32)
        return s[0] / n

    @staticmethod
    def backward(ctx, g):
        vc, faces32, pairs32 = ctx.saved_tensors
        n = pairs32.shape[0]
        d_v = L.zeros_like(vc)
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_normal_consistency_bwd(L.ptr(vc), L.ptr(faces32), L.ptr(pairs32), L.i32(n), L.ptr(gs), L.f32(1.0 / n), L.ptr(d_v),
                                                   L.stream()), 'normal_consistency_bwd')
        return d_v, None, None


def normal_consistency(v, faces32, pairs32):
    """mesh.py:18-28: mean (1 - cos(n_a, n_b))^2 over connected faces, n = cross(v1 - v0, v2 - v0) (un-normalised, mesh.py:242-254)"""
    return _NormalConsistencyFn.apply(v, faces32, pairs32)


class _CollisionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cloth, body, faces32, push_eps):
        cc, bc = cloth.contiguous().float(), body.contiguous().float()
        nc, nf = cc.shape[0], faces32.shape[0]
        centers = torch.empty(nf, 3, dtype=torch.float32, device=cc.device)
        L.check(L.lib().d3h_face_centers(L.ptr(bc), L.ptr(faces32), L.i32(nf), L.ptr(centers), L.stream()), 'face_centers')
        nn = _lbs.knn1(cc, centers)                      # knn_points(cloth, centres, K=1): squared L2, first minimum wins
        s = torch.empty(1, dtype=torch.float32, device=cc.device)
        L.check(L.lib().d3h_collision_fwd(L.ptr(cc), L.i32(nc), L.ptr(bc), L.ptr(faces32), L.ptr(nn), L.f32(push_eps), L.ptr(s), L.stream()),
                'collision_fwd')
        ctx.save_for_backward(cc, bc, faces32, nn)
        ctx.eps = float(push_eps)
        return s[0] / nc

    @staticmethod
    def backward(ctx, g):
        cc, bc, faces32, nn = ctx.saved_tensors
        nc = cc.shape[0]
        d_c = torch.empty_like(cc) if ctx.needs_input_grad[0] else None
        d_b = L.zeros_like(bc) if ctx.needs_input_grad[1] else None
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().d3h_collision_bwd(L.ptr(cc), L.i32(nc), L.ptr(bc), L.ptr(faces32), L.ptr(nn), L.f32(ctx.eps), L.ptr(gs), L.f32(1.0 / nc),
                                          L.ptr(d_c), L.ptr(d_b), L.stream()), 'collision_bwd')
        return d_c, d_b, None, None


def collision_loss(cloth_pos, body_pos, body_faces, push_eps=0.005):
    """geometry/hmsdf.py:98-132"""
    if body_faces.shape[0] == 3 and body_faces.ndim == 2 and body_faces.shape[1] != 3:
        body_faces = body_faces.T                                    # hmsdf.py:104-105
    f32 = body_faces if body_faces.dtype == torch.int32 else body_faces.int()
    return _CollisionFn.apply(cloth_pos, body_pos, f32.contiguous(), push_eps)


def mesh_sdf(points, verts, faces):
    """signed distance (positive OUTSIDE, negative inside) of points [n,3] to the closed, consistently wound triangle mesh
    (verts [V,3], faces [F,3]): what the SDF pre-fit gets as `-pysdf.SDF(verts, faces)(points)` in the reference (hmsdf.py:236-237)"""
    p = points.detach().contiguous().float()
    v = verts.detach().contiguous().float()
    f = faces if faces.dtype == torch.int32 else faces.int()
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    L.check(L.lib().d3h_mesh_sdf(L.ptr(p), L.i32(p.shape[0]), L.ptr(v), L.ptr(f.contiguous()), L.i32(f.shape[0]), L.ptr(out), L.stream()), 'mesh_sdf')
    return out
