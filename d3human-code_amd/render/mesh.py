"""Mesh container + auto_normals with the reference's interface (render/mesh.py:139-260,418-446).

Difference kept deliberately small but important for MI355X: the reference's constructor sorts + uniques all 3F edges on EVERY
construction (mesh.py:162,240-250; ~8 radix sorts per getMesh_* that init/split never read).  Here `edges` is computed on first
access with the same definition.  `t_pos_idx32` is the int32 view the HIP kernels consume."""
import torch

from d3h import imgops as _I


class Mesh:
    _FIELDS = ('v_pos', 't_pos_idx', 'v_nrm', 't_nrm_idx', 'v_tex', 't_tex_idx', 'v_tng', 't_tng_idx', 'material', 'kd', 'ks', 'uv',
               'uv_idx', 'face_labels', 'v_labels', 'connected_faces')

    def __init__(self, v_pos=None, t_pos_idx=None, v_nrm=None, t_nrm_idx=None, v_tex=None, t_tex_idx=None, v_tng=None, t_tng_idx=None,
                 material=None, base=None, kd=None, ks=None, uv=None, uv_idx=None, face_labels=None, v_labels=None, connected_faces=None,
                 edges=None, t_pos_idx32=None):
        loc = locals()
        for k in self._FIELDS:
            setattr(self, k, loc[k])
        self._edges = edges
        self._idx32 = t_pos_idx32
        if base is not None:
            self.copy_none(base)

    def copy_none(self, other):
        for k in self._FIELDS:
            if getattr(self, k) is None:
                setattr(self, k, getattr(other, k))
        if self._idx32 is None and self.t_pos_idx is other.t_pos_idx:
            self._idx32 = other._idx32
        if self._edges is None and self.t_pos_idx is other.t_pos_idx:
            self._edges = other._edges

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

    def get_edge(self):
        t = self.t_pos_idx
        e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
        return torch.unique(torch.sort(e, dim=1).values, dim=0)

    def clone(self):
        out = Mesh(base=self)
        for k in self._FIELDS:
            v = getattr(out, k)
            if torch.is_tensor(v):
                setattr(out, k, v.clone().detach())
        return out

    def getAABB(self):
        return torch.min(self.v_pos, dim=0).values, torch.max(self.v_pos, dim=0).values


def auto_normals(imesh):
    """mesh.py:418-446; v_pos may be [P,3] or, for a batch of posed frames, [B,P,3]"""
    v = imesh.v_pos
    f32 = imesh.t_pos_idx32
    v_nrm = _I.auto_normals(v, f32)             # one launch for the whole batch of posed frames
    return Mesh(v_nrm=v_nrm, t_nrm_idx=imesh.t_pos_idx, base=imesh)
