"""-m gpu: the HIP kernels through the C ABI against the oracle / the reference's golden vectors."""
import pytest

import parity_cases as PC

pytestmark = pytest.mark.gpu


def test_gpu_sdf_mlp_forward(gpu):
    PC.check_sdf_mlp_forward(gpu)


def test_gpu_marching_tets_golden(gpu):
    PC.check_mtets_golden(gpu)


def test_gpu_sdf_mlp_backward(gpu):
    PC.check_sdf_mlp_backward(gpu)
    PC.check_sdf_mlp_backward(gpu, n=1000, sparse_gout=True)
    PC.check_sdf_mlp_backward(gpu, n=777, sparse_gout=True)


def test_gpu_sdf_mlp_eikonal(gpu):
    PC.check_sdf_mlp_eikonal(gpu, n=4096)
    PC.check_sdf_mlp_eikonal(gpu, n=2999, scale=20.0)


def test_gpu_seq_ops(gpu):
    PC.check_seq_ops_golden(gpu)
    PC.check_mesh_api_seq(gpu)
    PC.check_mlp_deform_golden(gpu)


def test_gpu_sdf_mlp_deform(gpu):
    PC.check_sdf_mlp_deform(gpu)


def test_gpu_lbs_golden(gpu):
    PC.check_lbs_golden(gpu)


def test_gpu_rasterize(gpu):
    PC.check_rasterize(gpu, res=64)
    PC.check_rasterize(gpu, res=96, big=True, nb=1)


def test_gpu_interpolate(gpu):
    PC.check_interpolate(gpu, res=48)


def test_gpu_antialias(gpu):
    PC.check_antialias(gpu, res=48)


def test_gpu_texture(gpu):
    PC.check_texture(gpu)


def test_gpu_normals(gpu):
    PC.check_normals_golden(gpu)


def test_gpu_shading_normal(gpu):
    PC.check_shading_normal_golden(gpu)


def test_gpu_image_loss(gpu):
    PC.check_image_loss_golden(gpu)


def test_gpu_ssim(gpu):
    PC.check_ssim_golden(gpu)


def test_gpu_sample_points(gpu):
    PC.check_sample_points(gpu, nv=3000, nf=6000, n=50000)


def test_gpu_composite(gpu):
    PC.check_composite(gpu, B=3, H=67, W=129)


def test_gpu_pixel_losses(gpu):
    PC.check_pixel_losses(gpu, B=2, H=96, W=80)
    PC.check_pixel_losses(gpu, B=1, H=17, W=33, with_ssim=False)


def test_gpu_sdf_reg(gpu):
    PC.check_sdf_reg_golden(gpu)


def test_gpu_texmlp(gpu):
    PC.check_texmlp(gpu, n=5000)


def test_gpu_render_mesh_vs_reference_render(gpu):
    PC.check_render_mesh_golden(gpu)


def test_gpu_gshell_tangents(gpu):
    PC.check_gshell_tangents_golden(gpu)
