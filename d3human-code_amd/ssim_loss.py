"""ssim_loss.ssim with the reference's signature (ssim_loss.py:33-41) on the separable HIP kernels."""
from d3h.imgops import ssim  # noqa: F401


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()
