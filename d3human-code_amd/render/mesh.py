"""Mesh container + auto_normals with the reference's interface (render/mesh.py:139-260,418-446).

Difference kept deliberately small but important for MI355X: the reference's constructor sorts + uniques all 3F edges on EVERY
construction (mesh.py:162,240-250; ~8 radix sorts per getMesh_* that init/split never read).  Here `edges` is computed on first
access with the same definition.  `t_pos_idx32` is the int32 view the HIP kernels consume."""
import torch

from d3h import imgops as _I
from d3h import meshops as _M

find_edges = _M.find_edges                          # mesh.py:85-103
find_connected_faces = _M.find_connected_faces      # mesh.py:106-134 (vectorised; same pairs, same order)


def normal_consistency_loss(mesh):
    """mesh.py:18-28: mean (1 - cos)^2 between the (un-normalised) normals of faces sharing an edge"""
    return _M.normal_consistency(mesh.v_pos, mesh.t_pos_idx32, mesh.connected_faces32)


def compute_laplacian_uniform(mesh):
    """mesh.py:30-82: sparse [V,V] uniform Laplacian, L[i,j] = 1/deg(i) on edges, -1 on the diagonal.  The loss path
    (lap_loss.body_laplacian_loss) does not materialise it; this is the reference's accessor."""
    e = mesh.edges
    V = mesh.v_pos.shape[0]
    topo = _M.EdgeTopology.get(e, V)
    e0, e1 = e[:, 0], e[:, 1]
    idx = torch.cat([torch.stack([e0, e1]), torch.stack([e1, e0])], dim=1)
    val = torch.cat([topo.inv_deg[e0], topo.inv_deg[e1]])
    diag = torch.arange(V, device=e.device)
    idx = torch.cat([idx, torch.stack([diag, diag])], dim=1)
    val = torch.cat([val, -torch.ones(V, dtype=torch.float32, device=e.device)])
    return torch.sparse_coo_tensor(idx, val, (V, V)).coalesce()


class Mesh:
    _FIELDS = ('v_pos', 't_pos_idx', 'v_nrm', 't_nrm_idx', 'v_tex', 't_tex_idx', 'v_tng', 't_tng_idx', 'material', 'kd', 'ks', 'uv',
               'uv_idx', 'face_labels', 'v_labels', 'connected_faces')

    def __init__(self, v_pos=None, t_pos_idx=None, v_nrm=None, t_nrm_idx=None, v_tex=None, t_tex_idx=None, v_tng=None, t_tng_idx=None,
                 material=None, base=None, kd=None, ks=None, uv=None, uv_idx=None, face_labels=None, v_labels=None, connected_faces=None,
                 edges=None, t_pos_idx32=None):
        loc = locals()
        for k in self._FIELDS:
            setattr(self, k, loc[k])
        self._edges = None          # the reference overwrites a passed `edges` with get_edge() (mesh.py:158-162); here: on first access
        self._idx32 = t_pos_idx32
        self._v_nrm_lazy = None     # auto_normals(..., lazy=True): the normals are computed when somebody reads them
        if base is not None:
            self.copy_none(base)

    def copy_none(self, other):
        for k in self._FIELDS:
            if k == 'v_nrm':                    # do not force a deferred normal computation by copying it
                if self._v_nrm is None and self._v_nrm_lazy is None:
                    self._v_nrm, self._v_nrm_lazy = other._v_nrm, other._v_nrm_lazy
                continue
            if getattr(self, k) is None:
                setattr(self, k, getattr(other, k))
        if self._idx32 is None and self.t_pos_idx is other.t_pos_idx:
            self._idx32 = other._idx32
        if self._edges is None and self.t_pos_idx is other.t_pos_idx:
            self._edges = other._edges

    @property
    def v_nrm(self):
        if self._v_nrm is None and self._v_nrm_lazy is not None:
            self._v_nrm, self._v_nrm_lazy = self._v_nrm_lazy(), None
        return self._v_nrm

    @v_nrm.setter
    def v_nrm(self, v):
        self._v_nrm, self._v_nrm_lazy = v, None

    @property
    def t_pos_idx32(self):
        if self._idx32 is None:
            self._idx32 = self.t_pos_idx.int().contiguous()
        return self._idx32

    @property
    def edges(self):
        if self._edges is None:
            self._edges = self.get_edge()
        return self._edges

    @edges.setter
    def edges(self, v):
        self._edges = v

    @property
    def connected_faces32(self):
        if getattr(self, '_conn32', None) is None or self._conn32_src is not self.connected_faces:
            self._conn32_src = self.connected_faces
            self._conn32 = self.connected_faces.int().contiguous()
        return self._conn32

    def get_face_normals(self):
        """mesh.py:242-254: un-normalised cross(v1 - v0, v2 - v0)"""
        t, v = self.t_pos_idx, self.v_pos
        return torch.cross(v[t[:, 1]] - v[t[:, 0]], v[t[:, 2]] - v[t[:, 0]], dim=-1)

    @property
    def laplacian(self):
        return compute_laplacian_uniform(self)

    def normal_consistency(self):
        """mesh.py:266-279"""
        return normal_consistency_loss(self)

    _edge_cache = {}

    def get_edge(self):
        """mesh.py:240-250; memoised on the index tensor (fixed-topology stages build a Mesh over the same faces every iteration)"""
        t = self.t_pos_idx
        k = (t.data_ptr(), tuple(t.shape), t._version)
        hit = Mesh._edge_cache.get(k)
        if hit is not None and hit[0] is t:
            return hit[1]
        e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
        e = torch.unique(torch.sort(e, dim=1).values, dim=0)
        if len(Mesh._edge_cache) > 8:
            Mesh._edge_cache.clear()
        Mesh._edge_cache[k] = (t, e)
        return e

    def clone(self):
        out = Mesh(base=self)
        for k in self._FIELDS:
            v = getattr(out, k)
            if torch.is_tensor(v):
                setattr(out, k, v.clone().detach())
        return out

    def getAABB(self):
        return torch.min(self.v_pos, dim=0).values, torch.max(self.v_pos, dim=0).values


def auto_normals(imesh, lazy=False):
    """mesh.py:418-446; v_pos may be [P,3] or, for a batch of posed frames, [B,P,3].  lazy: defer the computation to the first read of
    `.v_nrm` (the canonical-space meshes of getMesh_* carry normals nobody reads in the training path)"""
    v = imesh.v_pos
    f32 = imesh.t_pos_idx32
    if lazy:
        m = Mesh(t_nrm_idx=imesh.t_pos_idx, base=imesh)
        m._v_nrm, m._v_nrm_lazy = None, (lambda: _I.auto_normals(v, f32))
        return m
    v_nrm = _I.auto_normals(v, f32)             # one launch for the whole batch of posed frames
    return Mesh(v_nrm=v_nrm, t_nrm_idx=imesh.t_pos_idx, base=imesh)
