"""SURVEY 8(f) rank 3 / 4 edges: target construction (dataset/dataset_split.py), checkpoint files (train.py:284-331,812-832), the
tet-grid file (geometry/hmsdf.py:207).  The expected values of the target builder are restated from the reference's lines in numpy."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
sys.path.insert(0, ROOT)


def test_target_builder_matches_reference_getitem():
    """golden = the reference's Dataset_split.__getitem__ / camera block run on in-memory images (tools/gen_golden.py: gen_data_edges)"""
    from dataset.targets import camera_matrices, get_ndc_matrix_from_ss, make_target
    from conftest import golden
    g = golden('data_edges.npz')
    cam = camera_matrices(g['K'], g['w2c'], 1080, 1080)
    assert np.array_equal(cam['proj'].numpy(), g['proj'])
    assert np.array_equal(cam['mv'].numpy(), g['mv']) and np.array_equal(cam['mvp'].numpy(), g['mvp'])
    assert np.array_equal(cam['campos'].numpy(), g['campos'])
    assert torch.equal(get_ndc_matrix_from_ss(540, 540, 600.0, 599.0, 270.0, 269.0), cam['proj'])
    H, W = g['rgb'].shape[:2]
    t = make_target(0, g['rgb'], g['msk'], g['cloth'], g['body'], g['nrm'], cam, [H, W], spp=1, device='cpu')
    for k in ('all_img', 'cloth_img', 'body_img', 'all_normal', 'body_normal', 'cloth_normal', 'all_msk', 'cloth_msk', 'body_msk', 'mv', 'mvp',
              'campos'):
        ref = g['t.' + k]
        assert tuple(t[k].shape) == ref.shape, k
        assert str(t[k].dtype).replace('torch.', '') == str(ref.dtype), (k, t[k].dtype, ref.dtype)
        assert np.array_equal(t[k].numpy(), ref), k
    assert set(np.unique(t['all_img'][0, ..., 3].numpy())) <= {0.0, 1.0}


def test_tet_grid_file_round_trip(tmp_path):
    from d3h import synth
    p = str(tmp_path / 'data' / 'tets' / 'tet_grid.npz')
    nv, nt = synth.write_tet_grid(p, 5)
    f = np.load(p)
    assert f['vertices'].dtype == np.float32 and f['indices'].dtype == np.int64 and f['vertices'].shape == (nv, 3) and f['indices'].shape == (nt, 4)
    v = torch.tensor(f['vertices'], dtype=torch.float32)      # the loader of geometry/hmsdf.py:207-211
    v[:, 1] = v[:, 1] - 0.1919
    v *= 1.2
    v_ref, t_ref = synth.kuhn_grid(5)
    assert np.array_equal(v.numpy(), v_ref) and np.array_equal(f['indices'], t_ref)


def test_checkpoint_files_round_trip(tmp_path):
    from d3h import checkpoint as C

    class Geo(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.sdf_net = torch.nn.Sequential(torch.nn.Linear(3, 4))
            self.msdf = torch.nn.Parameter(torch.rand(7))
            self.deform = torch.nn.Parameter(torch.zeros(7, 3))
    geo, tex = Geo(), torch.nn.Linear(2, 2)
    F = types.SimpleNamespace(init_epoch=501, **{k: torch.rand(2, 3) for k in C.POSE_KEYS})
    C.save_ckp(F, str(tmp_path / 'init'), 500, geo, {'kd_ks': tex})
    for name in ('model_500.pt', 'mtl_500.pt', 'smpl_500.pt.npz', 'probe_500.hdr'):                   # the reference's file names (train.py:815-832)
        assert os.path.exists(tmp_path / 'init' / 'ckp' / name)
    assert set(torch.load(tmp_path / 'init' / 'ckp' / 'model_500.pt').keys()) == {'sdf_net.0.weight', 'sdf_net.0.bias', 'msdf', 'deform'}
    want = {k: v.detach().clone() for k, v in geo.state_dict().items()}
    pose = F.trans_optim.clone()
    with torch.no_grad():
        for p in geo.parameters():
            p.add_(1.0)
    F.trans_optim = torch.zeros(2, 3)
    geo.msdf = torch.nn.Parameter(torch.rand(9))            # a shape that no longer matches is skipped (train.py:287), the rest loads
    C.load_ckp(F, str(tmp_path), geo, {'kd_ks': tex}, 'init')
    assert torch.equal(geo.deform, want['deform']) and torch.equal(geo.sdf_net[0].weight, want['sdf_net.0.weight']) and geo.msdf.shape == (9,)
    assert torch.equal(F.trans_optim, pose) and F.trans_optim.requires_grad
    # the probe is a readable Radiance picture of the constant 0.5 environment (train.py:1747); a missing pose tensor is refused
    raw = open(tmp_path / 'init' / 'ckp' / 'probe_500.hdr', 'rb').read()
    assert raw.startswith(b'#?RADIANCE') and b'-Y 16 +X 32' in raw
    assert np.array_equal(C.read_hdr(tmp_path / 'init' / 'ckp' / 'probe_500.hdr'), np.full((16, 32, 3), 0.5, np.float32))    # 0.5 = 128/256 * 2^0
    # a trained probe with a dark first pixel whose RGBE bytes are (2, 2, 1, *): flat scanlines would be mis-read as run-length coded
    probe = np.random.default_rng(0).random((4, 40, 3)).astype(np.float32)
    probe[:, 0] = np.array([2, 2, 1], np.float32) / 256.0 * 2.0 ** -5
    C.write_hdr(tmp_path / 'dark.hdr', probe)
    back = C.read_hdr(tmp_path / 'dark.hdr')
    assert np.abs(back - probe).max() <= np.abs(probe).max() / 128 and np.array_equal(back[:, 0], probe[:, 0])
    del F.jaw_pose_optim
    with pytest.raises(ValueError):
        C.save_ckp(F, str(tmp_path / 'init'), 501, geo, {'kd_ks': tex})
    assert len(set(np.load(tmp_path / 'init' / 'ckp' / 'smpl_500.pt.npz').files)) == 9
