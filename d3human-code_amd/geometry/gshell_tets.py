"""GShell_Tets with the reference's call signature (geometry/gshell_tets.py:253-447) on the sort-free HIP kernels.

Returns `(verts_aug, faces_aug, None, None, v_tng_aug, extra)` like the reference.  `v_tng_aug` / `extra['v_tng_watertight']`
(tangents from the uv atlas, :326-327,:380-386) are consumed by nothing in the training path (hmsdf.py:454-455 forwards only
verts/faces/uvs, and uvs are None): they are computed only when `self.compute_tangents = True` (a few library scatter ops on
the GPU), otherwise returned as None."""
import numpy as np
import torch

from d3h import mtets as _M
from d3h import imgops as _I


def _tangents(verts, faces, v_nrm, num_tets):
    """gshell_tets.py:40-83,219-248 with the reference's call (:327): uvs are looked up by VERTEX id (faces used as t_tex_idx)"""
    n = int(np.ceil(np.sqrt((num_tets * 2 + 1) // 2)))
    dev = verts.device
    lin = torch.linspace(0, 1 - (1 / n), n, dtype=torch.float32, device=dev)
    ty, tx = torch.meshgrid(lin, lin, indexing='ij')
    pad = 0.9 / n
    uvs = torch.stack([tx, ty, tx + pad, ty, tx + pad, ty + pad, tx, ty + pad], dim=-1).view(-1, 2)
    pos = [verts[faces[:, i]] for i in range(3)]
    tex = [uvs[faces[:, i]] for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]
    den = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]
    tang = nom / torch.where(den > 0.0, torch.clamp(den, min=1e-6), torch.clamp(den, max=-1e-6))
    tangents, tansum = torch.zeros_like(v_nrm), torch.zeros_like(v_nrm)
    for i in range(3):
        tangents = tangents.index_add(0, faces[:, i], tang)
        tansum = tansum.index_add(0, faces[:, i], torch.ones_like(tang))
    tangents = tangents / tansum
    sn = lambda x: x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=1e-20))
    tangents = sn(tangents)
    return sn(tangents - (tangents * v_nrm).sum(-1, keepdim=True) * v_nrm)


class GShell_Tets:
    compute_tangents = False

    def __call__(self, pos_nx3, sdf_n, msdf_n, tet_fx4, output_watertight_template=True, _body=False, _before_face_sync=None, _spec_hook=None):
        if not output_watertight_template:
            raise NotImplementedError('d3h GShell_Tets: output_watertight_template=False is never used by the reference')
        o = _M.marching_tets(pos_nx3, sdf_n, msdf_n, tet_fx4, body=_body, before_face_sync=_before_face_sync, spec_hook=_spec_hook)
        n_wt = o['n_wt']
        v_tng_aug = v_tng = None
        if self.compute_tangents and o['faces_wt'].shape[0] > 0:
            v_nrm = _I.auto_normals(o['verts_wt'], o['faces_wt32'])
            v_tng = _tangents(o['verts_wt'], o['faces_wt'], v_nrm, tet_fx4.shape[0])
            # boundary vertices: same interpolation weights as the positions (:342-386), from the watertight msdf values
            e = o['bnd_edge'].long()
            mv = o['msdf'][:n_wt]
            ma, mb = mv[e[:, 0]], mv[e[:, 1]]
            den = ma - mb
            ok = ((torch.sign(ma) + torch.sign(mb)).abs() != 2) & (den.abs() > 1e-12)
            den = torch.where(ok, den, torch.ones_like(den))
            w0 = torch.where(ok, -mb / den, torch.zeros_like(den))[:, None]
            w1 = torch.where(ok, ma / den, torch.zeros_like(den))[:, None]
            v_tng_aug = torch.cat([v_tng, v_tng[e[:, 0]] * w0 + v_tng[e[:, 1]] * w1], 0)
        extra = {'n_verts_watertight': n_wt, 'vertices_watertight': o['verts_wt'], 'faces_watertight': o['faces_wt'],
                 'v_tng_watertight': v_tng, 'msdf': o['msdf'], 'msdf_watertight': o['msdf'][:n_wt], 'msdf_boundary': o['msdf'][n_wt:],
                 'faces32': o['faces32'], 'faces_watertight32': o['faces_wt32']}
        return o['verts'], o['faces'], None, None, v_tng_aug, extra
