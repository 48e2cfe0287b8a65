"""The official SMPL-X model-file path (deform/smplx_exavatar_deformer.py:40 -> body_models.py:976-1078): `SMPLX_{GENDER}.npz` in the layout
of the licence-gated download -- V = 10 475, 55 joints, `shapedirs[..., 400]` (300 shape + 100 expression components), `posedirs`
[V,3,486], `kintree_table` uint32 with 2^32 - 1 as the root's parent, hand-PCA and landmark keys -- written synthetically by
d3h.synth.write_smplx_npz, loaded through `model_path=` exactly as the reference's constructor call does, and compared with the same
arrays handed over as `model_dict=` (what every other test uses)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def smplx_dir(tmp_path_factory):
    from d3h import synth
    root = tmp_path_factory.mktemp('model')
    d = synth.write_smplx_npz(str(root / 'smplx' / 'SMPLX_NEUTRAL.npz'), seed=3)
    return str(root), d


def _as_dict(d, n_shape=100, n_expr=50):
    return {'v_template': d['v_template'], 'weights': d['weights'], 'J_regressor': d['J_regressor'], 'shapedirs': d['shapedirs'][:, :, :n_shape],
            'expr_dirs': d['shapedirs'][:, :, 300:300 + n_expr], 'posedirs': d['posedirs'].reshape(-1, 486).T,
            'parents': np.where(d['kintree_table'][0].astype(np.int64) > 1000, -1, d['kintree_table'][0].astype(np.int64)), 'f': d['f'].astype(np.int64)}


def test_official_file_layout_loads_like_the_model_dict(smplx_dir):
    from deform.smplx_exavatar_deformer import SMPLX_Deformer
    root, d = smplx_dir
    assert d['v_template'].shape == (10475, 3) and d['shapedirs'].shape == (10475, 3, 400) and d['posedirs'].shape == (10475, 3, 486)
    assert d['kintree_table'].dtype == np.uint32 and int(d['kintree_table'][0, 0]) == 2 ** 32 - 1
    # the reference's call: SMPLX_Deformer(model_path='smplx', gender=FLAGS.gender) with the file at smplx/SMPLX_NEUTRAL.npz (hmsdf.py:191)
    a = SMPLX_Deformer(model_path=os.path.join(root, 'smplx'), gender='neutral', device='cpu')
    b = SMPLX_Deformer(model_path='unused', gender='neutral', model_dict=_as_dict(d), device='cpu')
    assert a.vertex_num == 10475 and a.face.shape == (20946, 3) and a.lbs_weights.shape == (10475, 55)
    assert int(a.layer.parents[0]) == -1 and torch.equal(a.layer.parents, b.layer.parents)
    for k in ('v_template', 'J_regressor', 'shapedirs', 'expr_dirs', 'posedirs', 'lbs_weights'):
        assert torch.equal(getattr(a.layer, k), getattr(b.layer, k)), k
    g = torch.Generator().manual_seed(0)
    betas = 0.5 * torch.randn(1, 100, generator=g)
    a.initialize(betas)
    b.initialize(betas)
    assert torch.equal(a.vs_template, b.vs_template) and torch.equal(a.init_A, b.init_A)
    assert a.vs_template.shape == (1, 10475, 3) and torch.isfinite(a.vs_template).all()
    # the same through the directory that CONTAINS smplx/ (model_path = the data root)
    c = SMPLX_Deformer(model_path=root, gender='neutral', device='cpu')
    assert torch.equal(c.layer.v_template, a.layer.v_template)
    with pytest.raises(FileNotFoundError):
        SMPLX_Deformer(model_path=os.path.join(root, 'smplx'), gender='female', device='cpu')


@pytest.mark.gpu
def test_gpu_reference_start_up_from_files_only(gpu, smplx_dir, tmp_path, monkeypatch):
    """the reference's start-up with NOTHING synthetic handed over in FLAGS: `smplx/SMPLX_NEUTRAL.npz` and `data/tets/tet_grid.npz` found
    relative to the working directory (hmsdf.py:191,207), the SDF pre-fit target computed from the template MESH of the file (hmsdf.py:232-237:
    pysdf there, the d3h_mesh_sdf kernel here), then one tick_init + backward"""
    import types
    from d3h import scene, synth
    from geometry.hmsdf import HmSDFTetsGeometry
    from render.mlptexture import MLPTexture3D
    import nvdiffrast.torch as dr
    root, d = smplx_dir
    monkeypatch.chdir(tmp_path)
    os.symlink(os.path.join(root, 'smplx'), tmp_path / 'smplx')
    synth.write_tet_grid(str(tmp_path / 'data' / 'tets' / 'tet_grid.npz'), 24)
    F = scene.make_flags(res=256, grid_n=2, n_frames=1, device='cuda', prefit_steps=200)
    F.smplx_model_dict, F.tet_grid, F.sdf_init_fn = None, None, None            # the file paths, the mesh-SDF pre-fit
    g = HmSDFTetsGeometry(48, 1.0, F)
    assert g.smplx_deform.vertex_num == 10475 and g.verts.shape[0] == 25 ** 3
    assert g.sdf_prefit_loss < 1e-3, g.sdf_prefit_loss                          # the network fitted the blob's signed distance
    t = lambda v: torch.tensor(v, dtype=torch.float32, device='cuda')
    mat = {'kd_ks': MLPTexture3D(g.getAABB(), channels=6, min_max=[torch.cat((t(F.kd_min)[0:3], t(F.ks_min))), torch.cat((t(F.kd_max)[0:3], t(F.ks_max)))]).cuda(),
           'bsdf': 'pbr'}
    mv, mvp, campos = synth.camera(256)
    H = 256
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing='ij')
    msk = ((((xx - 128) / 50) ** 2 + ((yy - 128) / 100) ** 2) < 1).float()[None, ..., None].cuda()
    tgt = {'idx': [0], 'mv': torch.from_numpy(mv)[None].cuda(), 'mvp': torch.from_numpy(mvp)[None].cuda(), 'campos': torch.from_numpy(campos)[None].cuda(),
           'resolution': [H, H], 'spp': 1, 'background': torch.rand(1, H, H, 3, device='cuda'),
           'all_img': torch.cat([0.5 * msk.expand(-1, -1, -1, 3), msk], -1).contiguous(), 'all_normal': torch.zeros(1, H, H, 3, device='cuda')}
    from render import renderutils as ru
    r = g.tick_init(dr.RasterizeGLContext(), tgt, None, mat, lambda a, b: ru.image_loss(a, b, loss='l1', tonemapper='log_srgb'), 0, None)
    total = r['reg_loss'] + r['normal_loss'] + r['msk_loss']
    total.backward()
    assert torch.isfinite(total) and g.last_mesh_dict['imesh'].t_pos_idx.shape[0] > 500
    assert float(r['msk_loss']) < 100 * 0.2                                      # the blob covers a plausible part of the ellipse target
    assert g.deform.grad is not None and torch.isfinite(g.deform.grad).all() and float(g.deform.grad.abs().max()) > 0
    assert F.trans_optim.grad is not None and torch.isfinite(F.trans_optim.grad).all()
