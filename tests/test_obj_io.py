"""render/obj.py: write_obj / write_ply / load_obj round trip with the reference's file layout."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))


def test_obj_ply_round_trip(tmp_path):
    from render import obj, mesh
    g = torch.Generator().manual_seed(0)
    v = torch.randn(7, 3, generator=g)
    f = torch.tensor([[0, 1, 2], [2, 3, 4], [4, 5, 6], [0, 2, 6]])
    n = torch.nn.functional.normalize(torch.randn(7, 3, generator=g), dim=-1)
    uv = torch.rand(5, 2, generator=g)
    fuv = torch.tensor([[0, 1, 2], [2, 3, 4], [4, 0, 1], [1, 2, 3]])
    m = mesh.Mesh(v, f, v_nrm=n, t_nrm_idx=f, v_tex=uv, t_tex_idx=fuv)
    path = obj.write_obj(str(tmp_path), m, save_material=False)
    text = open(path).read().splitlines()
    assert text[0] == 'mtllib mesh.mtl' and text[1] == 'g default' and 'usemtl defaultMat' in text
    assert sum(l.startswith('v ') for l in text) == 7 and sum(l.startswith('f ') for l in text) == 4
    assert text[-1].split()[1] == '1/2/1'                                     # 1-based v/vt/vn
    back = obj.load_obj(path, device='cpu')
    assert torch.allclose(back.v_pos, v) and torch.equal(back.t_pos_idx, f)
    assert torch.allclose(back.v_nrm, n) and torch.equal(back.t_nrm_idx, f)
    assert torch.allclose(back.v_tex, uv, atol=1e-6) and torch.equal(back.t_tex_idx, fuv)
    # positions only + quad fan triangulation
    m2 = mesh.Mesh(v, f)
    p2 = obj.write_obj(str(tmp_path), m2, save_name='plain.obj', save_material=False)
    assert open(p2).read().splitlines()[-1] == 'f  1// 3// 7//'
    with open(os.path.join(tmp_path, 'quad.obj'), 'w') as fh:
        fh.write('v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n')
    q = obj.load_obj(os.path.join(tmp_path, 'quad.obj'), device='cpu')
    assert q.t_pos_idx.tolist() == [[0, 1, 2], [0, 2, 3]] and q.v_nrm is None
    ply = obj.write_ply(str(tmp_path), mesh.Mesh(v, f, v_nrm=n, t_nrm_idx=f, v_tex=torch.rand(7, 2, generator=g), t_tex_idx=f))   # per-vertex uv
    lines = open(ply).read().splitlines()
    assert lines[0] == 'ply' and 'element vertex 7' in lines and 'element face 4' in lines and lines[-1] == '3 0 2 6'
    assert len(lines[lines.index('end_header') + 1].split()) == 8             # x y z nx ny nz s t
