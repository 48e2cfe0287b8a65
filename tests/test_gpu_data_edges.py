"""-m gpu: row f3 of SURVEY 8(f) on the device -- Dataset_split targets built on 'cuda' against the reference golden, a checkpoint of
GPU-resident geometry + material that reproduces the next training step after save -> load, OBJ / PLY export of an extracted GPU mesh."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gpu_dataset_split_targets_match_reference_golden(gpu):
    import test_data_edges as T
    ds = T.check_dataset_split_against_golden(gpu)
    t = ds[0]
    assert all(v.is_cuda for v in t.values() if torch.is_tensor(v))


def test_gpu_checkpoint_round_trip_reproduces_the_next_step(gpu, tmp_path):
    """save_ckp after a few steps (train.py:812-832 file set), perturb every parameter, load_ckp (train.py:292-331): the state and the
    losses of the next tick are those of the saved model"""
    from d3h.scene import Scene
    from d3h import checkpoint as C
    import e2e_cases as E
    sc = Scene(res=128, grid_n=12, n_frames=1, device=gpu, prefit_steps=80, loss_set='full', body_verts=1024,
               sdf_fn=lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4)
    for _ in range(3):
        sc.step()
    F, g, mat = sc.FLAGS, sc.geometry, sc.material
    F.init_epoch = 4
    C.save_ckp(F, str(tmp_path / 'init'), 3, g, mat)
    want = {k: v.detach().clone() for k, v in g.state_dict().items()}
    want_m = {k: v.detach().clone() for k, v in mat['kd_ks'].state_dict().items()}
    trans = F.trans_optim.detach().clone()

    def tick():
        torch.manual_seed(11)
        bg = torch.zeros(1, sc.res, sc.res, 3, device=gpu)
        pts = None
        r = g.tick_init(sc.glctx, sc.target(bg), None, mat, sc.loss_fn, 3, None)
        return {k: float(v) for k, v in r.items() if k in ('msk_loss', 'img_loss', 'sdf_reg_loss', 'normal_loss')}
    before = tick()
    with torch.no_grad():
        for p in list(g.parameters()) + list(mat['kd_ks'].parameters()):
            p.add_(0.05 * torch.randn_like(p))
        F.trans_optim = torch.zeros_like(F.trans_optim)
    close = lambda x, y: all(abs(x[k] - y[k]) <= 1e-5 * max(1e-6, abs(y[k])) for k in y)       # (float atomics: the sums are not bit-reproducible)
    assert not close(tick(), before)
    C.load_ckp(F, str(tmp_path), g, mat, 'init')
    assert all(torch.equal(v, want[k]) and v.is_cuda for k, v in g.state_dict().items())
    assert all(torch.equal(v, want_m[k]) for k, v in mat['kd_ks'].state_dict().items())
    assert torch.equal(F.trans_optim, trans) and F.trans_optim.is_cuda and F.trans_optim.requires_grad
    assert close(tick(), before)


def test_gpu_extracted_mesh_obj_ply_export(gpu, tmp_path):
    """render/obj.py write_obj / write_ply (obj.py:138,199; called by train.py:1004-1011,1359-1360) on the GPU-resident mesh of
    getMesh_init: positions, faces and normals round-trip through the file"""
    from d3h.scene import Scene
    from render import obj
    sc = Scene(res=128, grid_n=12, n_frames=1, device=gpu, prefit_steps=80, loss_set='mask', body_verts=1024,
               sdf_fn=lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4)
    with torch.no_grad():
        m = sc.geometry.getMesh_init(sc.material)['imesh']
    assert m.v_pos.is_cuda and m.t_pos_idx.shape[0] > 0
    p = obj.write_obj(str(tmp_path), m, save_material=False)
    back = obj.load_obj(p, device='cpu')
    assert torch.allclose(back.v_pos, m.v_pos.cpu(), atol=1e-5) and torch.equal(back.t_pos_idx, m.t_pos_idx.cpu())
    ply = obj.write_ply(str(tmp_path), m)
    lines = open(ply).read().splitlines()
    assert f'element vertex {m.v_pos.shape[0]}' in lines and f'element face {m.t_pos_idx.shape[0]}' in lines
