"""Kernel-logic checks on the host emulation of the .hip sources (toy sizes; see tests/emul/hip_emul.h).

These do NOT replace the -m gpu parity tests: they exist so index maps / MFMA fragment layouts /
compaction order are debugged before a GPU round trip.
"""
import parity_cases as PC


def test_emul_sdf_mlp_forward(emul):
    PC.check_sdf_mlp_forward(emul, n=200)


def test_emul_marching_tets_golden(emul):
    PC.check_mtets_golden(emul)


def test_emul_sdf_mlp_backward(emul):
    PC.check_sdf_mlp_backward(emul, n=96)


def test_emul_lbs_golden(emul):
    PC.check_lbs_golden(emul)


def test_emul_rasterize(emul):
    PC.check_rasterize(emul, res=32)
    PC.check_rasterize(emul, res=40, big=True, nb=1)


def test_emul_interpolate(emul):
    PC.check_interpolate(emul, res=32)


def test_emul_antialias(emul):
    PC.check_antialias(emul, res=32)


def test_emul_texture(emul):
    PC.check_texture(emul)


def test_emul_image_ops(emul):
    PC.check_normals_golden(emul)
    PC.check_shading_normal_golden(emul)
    PC.check_image_loss_golden(emul)
    PC.check_ssim_golden(emul)
    PC.check_sdf_reg_golden(emul)


def test_emul_texmlp(emul):
    PC.check_texmlp(emul, n=300)


def test_emul_render_mesh_vs_reference_render(emul):
    PC.check_render_mesh_golden(emul)
