"""The CPU oracle against the golden vectors produced by the reference itself (tools/gen_golden.py)."""
import glob
import os

import numpy as np
import torch

from conftest import GOLD, golden
from oracle import marching_tets as OMT
from oracle import sdf_mlp as OMLP


def test_oracle_sdf_mlp_matches_reference():
    g = golden('sdf_mlp.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = OMLP.mlp_forward(x, sd)
    assert (y.detach().numpy() - g['sdf']).__abs__().max() < 1e-7
    (y * torch.from_numpy(g['gout'])).sum().backward()
    assert np.abs(x.grad.numpy() - g['dx']).max() < 1e-5 * np.abs(g['dx']).max() + 1e-7


def test_oracle_marching_tets_matches_reference():
    for f in sorted(glob.glob(os.path.join(GOLD, 'mtets_*.npz'))):
        g = np.load(f)
        o = OMT.gshell_tets(torch.from_numpy(g['in_pos']), torch.from_numpy(g['in_sdf']), torch.from_numpy(g['in_msdf']),
                            torch.from_numpy(g['tets']), negate_msdf=('body' in f))
        assert np.array_equal(o['faces'].numpy(), g['faces']), f
        assert np.array_equal(o['faces_watertight'].numpy(), g['faces_watertight']), f
        for k in ('verts', 'v_tng', 'vertices_watertight', 'msdf', 'msdf_watertight', 'msdf_boundary'):
            if g[k].size:
                assert np.abs(o[k].detach().numpy() - g[k]).max() < 1e-6, (f, k)


def test_kuhn_grid_sizes():
    v, t = OMT.kuhn_grid(4)
    assert v.shape == (125, 3) and t.shape == (384, 4)
    # every tet has positive volume magnitude (non-degenerate)
    p = v[t]
    vol = np.abs(np.einsum('ij,ij->i', np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0]))
    assert vol.min() > 1e-6


def test_oracle_seq_terms_match_reference():
    """oracle/seq_ops.py against the outputs of the reference's own collision_loss / Mesh.laplacian / normal_consistency /
    find_connected_faces / MLP_deform (tests/golden/seq.npz), values and gradients"""
    from oracle import seq_ops as OS
    g = np.load(os.path.join(GOLD, 'seq.npz'))
    T = lambda k, grad=False: torch.from_numpy(g[k]).requires_grad_(grad)
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    c, b = T('cloth_v', True), T('body_v', True)
    l = OS.collision_loss(c, b, T('body_f'), float(g['colli_eps']))
    assert abs(l.item() - float(g['colli'])) < 1e-6 * float(g['colli'])
    l.backward()
    assert rel(c.grad.numpy(), g['colli_dcloth']) < 1e-5 and rel(b.grad.numpy(), g['colli_dbody']) < 1e-5
    pairs, _ = OS.find_connected_faces(T('all_f'))
    assert np.array_equal(pairs.numpy(), g['connected_faces'])
    assert np.array_equal(OS.find_edges(T('all_f')).numpy(), g['edges_unique'])
    v = T('all_v', True)
    ll = OS.laplacian_uniform_loss(v, T('mesh_edges'))
    ll.backward()
    assert abs(ll.item() - float(g['lap'])) < 1e-6 * float(g['lap']) and rel(v.grad.numpy(), g['lap_dv']) < 1e-5
    v2 = T('all_v', True)
    nl = OS.normal_consistency_loss(v2, T('all_f'), pairs)
    nl.backward()
    assert abs(nl.item() - float(g['ncons'])) < 1e-5 * float(g['ncons']) and rel(v2.grad.numpy(), g['ncons_dv']) < 1e-4
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('nr_sd.')}
    y = OS.mlp_deform_forward(T('nr_x'), T('nr_code'), sd, n_freq=8, skip_layers=tuple(int(i) for i in g['nr_skip_layers']))
    assert (y - T('nr_y')).abs().max() < 1e-6


def test_oracle_image_ops_match_reference():
    """oracle/image_ops.py against the reference's auto_normals / prepare_shading_normal / image_loss / ssim / sdf_reg outputs
    (tests/golden/imgops.npz)"""
    from oracle import image_ops as OI
    g = np.load(os.path.join(GOLD, 'imgops.npz'))
    T = lambda k: torch.from_numpy(g[k])
    assert (OI.auto_normals(T('an_v'), T('an_f')) - T('an_out')).abs().max() < 1e-6
    for tag, two_sided in (('psn2_', True), ('psn1_', False)):
        o = OI.prepare_shading_normal(T(tag + 'in_pos'), T(tag + 'in_view'), T(tag + 'in_pert'), T(tag + 'in_snrm'), T(tag + 'in_stng'),
                                      T(tag + 'in_gnrm'), two_sided, True)
        assert (o - T(tag + 'out')).abs().max() < 1e-6, tag
    for loss in ('l1', 'mse', 'smape', 'relmse'):
        assert abs(OI.image_loss(T('il_a'), T('il_b'), loss).item() - float(g[f'il_{loss}'])) < 1e-6, loss
    assert abs(OI.ssim(T('ssim_x'), T('ssim_y')).item() - float(g['ssim'])) < 1e-6
    assert abs(OI.sdf_reg_loss(T('reg_sdf'), T('reg_edges')).item() - float(g['reg'])) < 1e-6


def test_oracle_lbs_matches_reference():
    """oracle/lbs.py against the reference's lbs() / SMPLX_Deformer outputs on the seeded miniature model (tests/golden/lbs.npz)"""
    from oracle import lbs as OL
    g = np.load(os.path.join(GOLD, 'lbs.npz'))
    T = lambda k: torch.from_numpy(g[k])
    o, idx, cano = OL.lbs_forward(T('pts'), T('tmpl'), T('model.weights'), T('A0'), T('A')[0], T('trans')[0])
    assert (o - T('out')[0]).abs().max() < 2e-6 and (cano - T('canonical')).abs().max() < 2e-6
    _, w = OL.nearest_weights(T('pts'), T('tmpl'), T('model.weights'))
    assert torch.equal(w, T('w_pts'))
    for f in range(T('A').shape[0]):
        of, _, _ = OL.lbs_forward(T('pts'), T('tmpl'), T('model.weights'), T('A0'), T('A')[f], T('trans')[f])
        assert (of - T('out')[f]).abs().max() < 2e-6, f
