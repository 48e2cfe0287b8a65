"""render.renderutils with the reference's entry points (render/renderutils/ops.py:197,479,518,541) on the HIP kernels.
The BSDF / cubemap functions of the reference plugin are dead under the hard-wired bsdf='kd' (render/render.py:120) and are
not provided."""
import torch

from d3h import imgops as _I

__all__ = ['xfm_points', 'xfm_vectors', 'image_loss', 'prepare_shading_normal']


def xfm_points(points, matrix, use_python=True):
    """[B or 1,V,3] x [B,4,4] -> [B,V,4] homogeneous (ops.py:518-537).  One thread per point (d3h_xfm_points_*) when the matrix is a
    constant [B,4,4] and the points are [B or 1,V,3]; the reference's matmul formulation otherwise."""
    if (points.dim() == 3 and matrix.dim() == 3 and not matrix.requires_grad and points.shape[0] in (1, matrix.shape[0])
            and points.shape[-1] == 3 and points.dtype == torch.float32):
        return _I.xfm_points(points, matrix, 1.0)
    return torch.matmul(torch.nn.functional.pad(points, pad=(0, 1), mode='constant', value=1.0), torch.transpose(matrix, 1, 2))


def xfm_vectors(vectors, matrix, use_python=True):
    return torch.matmul(torch.nn.functional.pad(vectors, pad=(0, 1), mode='constant', value=0.0), torch.transpose(matrix, 1, 2))[..., 0:3].contiguous()


def image_loss(img, target, loss='l1', tonemapper='none', use_python=False):
    return _I.image_loss(img, target, loss=loss, tonemapper=tonemapper)


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True, use_python=False):
    return _I.prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl)
