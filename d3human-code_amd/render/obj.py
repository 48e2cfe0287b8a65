"""render/obj.py of the reference (load_obj :31-133, write_ply :138-197, write_obj :199-256): Wavefront OBJ / ASCII PLY mesh IO with the
same file layout (`mtllib mesh.mtl`, groups, `f v/vt/vn` with 1-based indices, vt written as (u, 1 - v)).  Host-side data edge of the
path: plain numpy string formatting, vectorised instead of one Python write per element."""
import os

import numpy as np
import torch

from . import mesh


def _np(t):
    return t.detach().cpu().numpy() if t is not None else None


def write_obj(folder, mesh, save_name=None, save_material=True):
    obj_file = os.path.join(folder, 'mesh.obj' if save_name is None else save_name)
    v_pos, v_nrm, v_tex = _np(mesh.v_pos), _np(mesh.v_nrm), _np(mesh.v_tex)
    t_pos, t_nrm, t_tex = _np(mesh.t_pos_idx), _np(mesh.t_nrm_idx), _np(mesh.t_tex_idx)
    if v_pos.ndim == 3:                  # a batch of posed frames: the first one
        v_pos = v_pos[0]
        v_nrm = v_nrm[0] if v_nrm is not None and v_nrm.ndim == 3 else v_nrm
    lines = ['mtllib mesh.mtl', 'g default']
    lines += ['v {} {} {} '.format(*v) for v in v_pos.tolist()]
    if v_tex is not None:
        assert len(t_pos) == len(t_tex)
        lines += ['vt {} {} '.format(v[0], 1.0 - v[1]) for v in v_tex.tolist()]
    if v_nrm is not None:
        assert len(t_pos) == len(t_nrm)
        lines += ['vn {} {} {}'.format(*v) for v in v_nrm.tolist()]
    lines += ['s 1 ', 'g pMesh1', 'usemtl defaultMat']
    for i in range(len(t_pos)):
        lines.append('f ' + ''.join(' %s/%s/%s' % (t_pos[i][j] + 1, '' if v_tex is None else t_tex[i][j] + 1,
                                                   '' if v_nrm is None else t_nrm[i][j] + 1) for j in range(3)))
    with open(obj_file, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    if save_material and mesh.material is not None and hasattr(mesh.material, 'get') and mesh.material.get('kd') is not None:
        from . import material as _material            # texture-map materials only; MLP materials have no .mtl form
        _material.save_mtl(os.path.join(folder, 'mesh.mtl'), mesh.material)
    return obj_file


def write_ply(folder, mesh, save_name=None):
    ply_file = os.path.join(folder, 'mesh.ply' if save_name is None else save_name)
    v_pos, v_nrm, v_tex, t_pos = _np(mesh.v_pos), _np(mesh.v_nrm), _np(mesh.v_tex), _np(mesh.t_pos_idx)
    if v_pos.ndim == 3:
        v_pos = v_pos[0]
        v_nrm = v_nrm[0] if v_nrm is not None and v_nrm.ndim == 3 else v_nrm
    head = ['ply', 'format ascii 1.0', 'element vertex {}'.format(len(v_pos)), 'property float x', 'property float y', 'property float z']
    cols = [v_pos]
    if v_nrm is not None:
        head += ['property float nx', 'property float ny', 'property float nz']
        cols.append(v_nrm)
    if v_tex is not None:
        head += ['property float s', 'property float t']
        cols.append(v_tex)
    head += ['element face {}'.format(len(t_pos)), 'property list uchar int vertex_indices', 'end_header']
    rows = np.concatenate(cols, axis=1).tolist()
    with open(ply_file, 'w') as f:
        f.write('\n'.join(head) + '\n')
        f.write('\n'.join(' '.join(str(x) for x in r) for r in rows) + '\n')
        f.write('\n'.join('3 {} {} {}'.format(*t) for t in t_pos.tolist()) + '\n')
    return ply_file


def load_obj(filename, clear_ks=True, mtl_override=None, mtl_default=None, mtl_type_override=None, device=None):
    """positions / texcoords / normals / triangulated faces of one OBJ (obj.py:31-133 without its material-merging branch: the mesh
    comes back with material=mtl_default)."""
    v, vt, vn, f, ft, fn = [], [], [], [], [], []
    with open(filename) as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            tag = p[0].lower()
            if tag == 'v':
                v.append([float(x) for x in p[1:4]])
            elif tag == 'vt':
                vt.append([float(p[1]), 1.0 - float(p[2])])
            elif tag == 'vn':
                vn.append([float(x) for x in p[1:4]])
            elif tag == 'f':
                idx = [(q.split('/') + ['', ''])[:3] for q in p[1:]]
                conv = lambda s: int(s) - 1 if s != '' else -1
                for k in range(1, len(idx) - 1):              # fan triangulation (obj.py:100-121)
                    tri = (idx[0], idx[k], idx[k + 1])
                    f.append([conv(t[0]) for t in tri])
                    ft.append([conv(t[1]) for t in tri])
                    fn.append([conv(t[2]) for t in tri])
    dev = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
    T = lambda a, dt: torch.tensor(a, dtype=dt, device=dev) if len(a) else None
    return mesh.Mesh(T(v, torch.float32), T(f, torch.int64), T(vn, torch.float32), T(fn, torch.int64) if vn else None, T(vt, torch.float32),
                     T(ft, torch.int64) if vt else None, material=mtl_default)
