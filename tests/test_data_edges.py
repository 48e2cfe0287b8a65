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


def _memory_source(g, n_frames=3):
    """the golden's images as a 3-frame in-memory sequence (frame 1 = the golden frame), per-frame SMPL-X rows, the golden camera"""
    from dataset.dataset_split import MemorySource, SMPLX_KEYS, _SMPLX_WIDTH
    rng = np.random.default_rng(3)
    frames = []
    for k in range(n_frames):
        if k == 1:
            frames.append((g['rgb'], g['msk'], g['cloth'], g['body'], g['nrm']))
        else:
            frames.append((rng.integers(0, 256, g['rgb'].shape, dtype=np.uint8), g['msk'], g['cloth'], g['body'], g['nrm']))
    smplx = {k: rng.standard_normal((n_frames, _SMPLX_WIDTH[k])).astype(np.float32) for k in SMPLX_KEYS}
    smplx.update(face_offset=np.zeros((1, 8, 3), np.float32), joint_offset=np.zeros((1, 55, 3), np.float32),
                 locator_offset=np.zeros((1, 55, 3), np.float32), shape_param=np.zeros((1, 100), np.float32))
    cam = {'intrinsic': g['K'], 'extrinsic': g['w2c'], 'height': 1080, 'width': 1080}
    return MemorySource(frames, (0, n_frames - 1), smplx, cam), smplx


TARGET_KEYS = ('all_img', 'cloth_img', 'body_img', 'all_normal', 'body_normal', 'cloth_normal', 'all_msk', 'cloth_msk', 'body_msk', 'mv', 'mvp', 'campos')


def check_dataset_split_against_golden(device):
    """Dataset_split.__getitem__ / collate (dataset/dataset_split.py:109-283, dataset/dataset.py:146-189) on an in-memory sequence against
    the reference's own __getitem__ (tests/golden/data_edges.npz), tensors on `device`"""
    from dataset.dataset_split import Dataset_split
    from conftest import golden
    g = golden('data_edges.npz')
    H, W = g['rgb'].shape[:2]
    src, smplx = _memory_source(g)
    F = types.SimpleNamespace(train_res=[W, H], spp=1, device=device)
    ds = Dataset_split(src, F)
    assert len(ds) == 2 and ds.key_frame == [0, 1, 2] and ds.begin == 0 and ds.end == 2          # n_images = end - begin (:138)
    assert len(Dataset_split(src, F, examples=7)) == 7
    t = ds[1]
    assert t['idx'] == 1 and t['resolution'] == [W, H] and t['spp'] == 1
    for k in TARGET_KEYS:
        ref = g['t.' + k]
        assert t[k].device.type == torch.device(device).type, k
        assert tuple(t[k].shape) == ref.shape and str(t[k].dtype).replace('torch.', '') == str(ref.dtype), (k, t[k].shape, t[k].dtype)
        assert np.array_equal(t[k].cpu().numpy(), ref), k
    for k in ('trans', 'rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose'):
        assert np.array_equal(t[k].cpu().numpy(), smplx[k][1][None]), k
    assert ds[3]['idx'] == 1                                                                     # itr % n_images (:207)
    b = ds.collate([ds[0], ds[1]])
    assert b['idx'] == [0, 1] and b['all_img'].shape == (2, H, W, 4) and b['mvp'].shape == (2, 4, 4) and b['body_pose'].shape == (2, 63)
    assert b['img'] is None and b['normal'] is None and np.array_equal(b['cloth_normal'][1].cpu().numpy(), g['t.cloth_normal'][0])
    return ds


def test_dataset_split_matches_reference_getitem():
    check_dataset_split_against_golden('cpu')


def test_dataset_split_reads_the_reference_directory_layout(tmp_path):
    """the on-disk layout of dataset_split.py:115-136,161-163 (PNG files written with PIL, key.list, smplx/*.npz, *.json, cameras.npz)
    gives the same targets as the in-memory source; with Detail=True the merged body + garment mesh files of :149-163 are read too"""
    import json
    from PIL import Image
    from dataset.dataset_split import Dataset_split, SMPLX_KEYS
    from conftest import golden
    g = golden('data_edges.npz')
    H, W = g['rgb'].shape[:2]
    src, smplx = _memory_source(g)
    base = tmp_path / 'seq'
    for sub in ('images', 'normal', 'all', 'all_cloth_mask', 'all_body_mask', 'smplx/smplx_optimized', 'proc'):
        os.makedirs(base / sub)
    for k, (rgb, msk, cloth, body, nrm) in enumerate(src.frames):
        for sub, a in (('images', rgb), ('normal', nrm), ('all', msk), ('all_cloth_mask', cloth), ('all_body_mask', body)):
            Image.fromarray(np.ascontiguousarray(a)).save(base / sub / f'{k:04d}.png')
    (base / 'key.list').write_text('0\n2\n')
    np.savez(base / 'smplx' / 'merged_smplx.npz', **{k: smplx[k] for k in SMPLX_KEYS})
    for k in ('face_offset', 'joint_offset', 'locator_offset', 'shape_param'):
        (base / 'smplx' / 'smplx_optimized' / f'{k}.json').write_text(json.dumps(smplx[k][0].tolist()))
    np.savez(base / 'smplx' / 'cameras.npz', intrinsic=g['K'], extrinsic=g['w2c'], height=1080, width=1080)
    v = np.random.default_rng(1).standard_normal((6, 3)).astype(np.float32)
    f = np.array([[0, 1, 2], [2, 3, 4], [3, 4, 5]], np.int64)
    np.savez(base / 'proc' / 'merge_body_cloth.npz', v=v, f=f, face_labels=np.array([0, 1, 1]))
    np.savez(base / 'proc' / 'inside_body_index.npz', inside_body_index=np.array([0]), outside_body_index=np.array([1]))
    F = types.SimpleNamespace(train_res=[W, H], spp=1, device='cpu')
    ds = Dataset_split(str(base), F, Detail=True, process_path=str(base / 'proc'))
    t = ds[1]
    for k in TARGET_KEYS:
        assert np.array_equal(t[k].numpy(), g['t.' + k]), k
    assert ds.shape_param.shape == (1, 100) and ds.joint_offset.shape == (1, 55, 3)
    assert ds.cloth_index.tolist() == [2, 3, 4, 5] and ds.outside_index.tolist() == [2, 3, 4, 5, 1] and ds.v.dtype == torch.float32
    # a sequence stored at another resolution is resized to FLAGS.train_res (bilinear), masks stay binary
    F2 = types.SimpleNamespace(train_res=[W // 2, H // 2], spp=1, device='cpu')
    t2 = Dataset_split(str(base), F2)[1]
    assert t2['all_img'].shape == (1, H // 2, W // 2, 4) and set(np.unique(t2['all_msk'].numpy())) <= {0.0, 1.0}


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


def test_normal_png_decoding_is_always_three_by_eight_bit(tmp_path):
    """DirectorySource reads the normal maps as cv2.imread(IMREAD_COLOR) does in the reference (dataset_split.py:248-249): H x W x 3 uint8
    whatever the file holds -- grey, palette, RGBA, 16-bit (`[..., :3]` on an H x W grey image sliced the WIDTH)"""
    from PIL import Image
    from dataset.dataset_split import _read_color_png
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (5, 7), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'rgb.png')
    Image.fromarray(np.concatenate([rgb, np.full((5, 7, 1), 200, np.uint8)], -1)).save(tmp_path / 'rgba.png')
    Image.fromarray(grey).save(tmp_path / 'grey.png')
    Image.fromarray(rgb).convert('P', palette=Image.ADAPTIVE, colors=256).save(tmp_path / 'pal.png')
    Image.fromarray((grey.astype(np.uint16) << 8) | 0x34).save(tmp_path / 'g16.png')
    for name in ('rgb', 'rgba'):
        assert np.array_equal(_read_color_png(str(tmp_path / f'{name}.png')), rgb)
    for name in ('grey', 'g16'):
        a = _read_color_png(str(tmp_path / f'{name}.png'))
        assert a.shape == (5, 7, 3) and a.dtype == np.uint8 and np.array_equal(a, np.stack([grey] * 3, -1))
    p = _read_color_png(str(tmp_path / 'pal.png'))
    assert p.shape == (5, 7, 3) and p.dtype == np.uint8 and np.abs(p.astype(int) - rgb.astype(int)).max() <= 8      # (35 colours fit the palette)
