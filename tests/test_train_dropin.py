"""The drop-in boundary, executed: the REFERENCE's own train.py (`optimize_mesh_init` train.py:544-832, `optimize_mesh_split` :839-1243,
`optimize_mesh_seq` :1246-1525) imported with this build first on the module path and driven for a few iterations.  Needs /root/reference (dev container only -- it cannot travel to
the GPU box), so the kernels run on the host emulator and device='cuda' literals are rewritten to 'cpu' (tools/run_reference_train.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'train.py')), reason='the reference checkout is not present on this machine')
def test_reference_train_py_runs_unchanged_on_this_build(emul_lib, tmp_path):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'd3human-code_amd') + os.pathsep + REF)
    p = subprocess.run([sys.executable, '-u', os.path.join(ROOT, 'tools', 'run_reference_train.py'), '--emulator', '--iters', '1', '--grid', '5',
                        '--res', '24', '--eik', '32', '--out', str(tmp_path)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "'train': '/root/reference/train.py'" in out and "'geometry.hmsdf': 'd3human-code_amd/geometry/hmsdf.py'" in out
    assert "'dataset.dataset_split': 'd3human-code_amd/dataset/dataset_split.py'" in out         # train.py:25 gets the build's Dataset_split
    assert "'render.util': 'd3human-code_amd/render/util.py'" in out
    lines = [l for l in out.splitlines() if l.startswith('iter=')]
    assert len(lines) == 2 and all('nan' not in l for l in lines), lines
    assert p.stdout.strip().endswith('OK')


def _drive(stage, tmp_path, iters):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'd3human-code_amd') + os.pathsep + REF)
    p = subprocess.run([sys.executable, '-u', os.path.join(ROOT, 'tools', 'run_reference_train.py'), '--emulator', '--stage', stage, '--iters', str(iters),
                        '--grid', '5', '--res', '24', '--eik', '32', '--out', str(tmp_path)], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=1500)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "'train': '/root/reference/train.py'" in out and "'geometry.hmsdf': 'd3human-code_amd/geometry/hmsdf.py'" in out
    assert p.stdout.strip().endswith('OK')
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'train.py')), reason='the reference checkout is not present on this machine')
def test_reference_optimize_mesh_split_runs_unchanged_on_this_build(emul_lib, tmp_path):
    """train.py:839-1243: prepare_batch_split, validate_itr_all (render_split x2, all 12 buffers) at iteration 0, tick_split for the garment
    and the body per iteration, the total of :1087, backward, encoder gradient / 8, both Adam optimisers + schedulers, clamp_deform"""
    out = _drive('split', tmp_path, 1)
    lines = [l for l in out.splitlines() if l.startswith('iter=')]
    assert len(lines) == 2 and all('nan' not in l for l in lines) and all('cloth_msk_loss=' in l and 'body_normal_loss=' in l for l in lines), lines
    assert 'optimize_mesh_split returned tuple' in out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'train.py')), reason='the reference checkout is not present on this machine')
def test_reference_optimize_mesh_seq_runs_unchanged_on_this_build(emul_lib, tmp_path):
    """train.py:1246-1525: tick_seq per iteration, the total of :1412-1421, backward, both optimisers; at the last iteration
    validate_all_mesh (render_seq with every buffer), write_ply x2, the delta / visible-triangle dump (the lazy visible_triangles of this
    build goes through `.detach().cpu().numpy()` there)"""
    out = _drive('seq', tmp_path, 1)
    lines = [l for l in out.splitlines() if l.startswith('iter=')]
    assert len(lines) == 2 and all('nan' not in l for l in lines) and 'times=  299' in lines[-1], lines
    assert 'optimize_mesh_seq returned tuple' in out


def test_render_util_covers_the_reference_surface():
    """every public function of the reference's render/util.py exists in the build's (train.py, light.py, material.py, texture.py and the
    denoiser import `render.util` and get the build's module); the list is data (names only), checked without the reference present"""
    sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
    import importlib.util
    spec = importlib.util.spec_from_file_location('_d3h_render_util', os.path.join(ROOT, 'd3human-code_amd', 'render', 'util.py'))
    u = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(u)
    names = ('dot reflect length safe_normalize to_hvec ycocg2rgb hsv2rgb pixel_grid dilate rgb_to_srgb srgb_to_rgb reinhard mse_to_psnr '
             'psnr_to_mse get_miplevels tex_2d cube_to_dir latlong_to_cubemap cubemap_to_latlong scale_img_hwc scale_img_nhwc avg_pool_nhwc '
             'segment_sum fovx_to_fovy focal_length_to_fovy perspective perspective_offcenter translate rotate_x rotate_y rotate_z scale lookAt '
             'random_rotation_translation random_rotation lines_focal cosine_sample bilinear_downsample display_image save_image '
             'save_image_raw load_image_raw load_image time_to_text checkerboard').split()
    missing = [n for n in names if not callable(getattr(u, n, None))]
    assert not missing, missing
    import numpy as np
    import torch
    c = u.checkerboard((5, 6), 2)
    assert c.shape == (5, 6, 3) and abs(c[0, 0, 0] - 0.66) < 1e-9 and abs(c[0, 2, 0] - 0.33) < 1e-9
    assert u.time_to_text(30) == '30.00 s' and u.time_to_text(90) == '1.50 m' and u.time_to_text(7200) == '2.00 h'
    assert torch.allclose(u.rotate_z(0.3) @ u.rotate_z(-0.3), torch.eye(4), atol=1e-6)
    x, m = torch.rand(1, 8, 8, 3), torch.zeros(1, 8, 8, 1)
    m[:, 2:6, 2:6] = 1
    d = u.dilate(x, torch.zeros(1, 1, 1, 3), m, 5)
    assert torch.equal(d * m, x * m) and float((d * (1 - m)).abs().sum()) > 0
