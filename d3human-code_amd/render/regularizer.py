"""Shading / material regularisers used by tick_split (render/regularizer.py:22-52): chroma_loss, material_smoothness_grad.
Small elementwise reductions over the already-rendered buffers (plumbing-level torch ops on the GPU)."""
import torch


def _value3(x):
    return torch.max(x[..., 0:3], dim=-1, keepdim=True)[0].expand(-1, -1, -1, 3)


def luma(x):
    return ((x[..., 0:1] + x[..., 1:2] + x[..., 2:3]) / 3).expand(-1, -1, -1, 3)


def value(x):
    return _value3(x)


def chroma_loss(kd, color_ref, lambda_chroma):
    """regularizer.py:22-26"""
    eps = 0.001
    ref = color_ref[..., 0:3] / torch.clip(_value3(color_ref), min=eps)
    opt = kd[..., 0:3] / torch.clip(_value3(kd), min=eps)
    return torch.mean(torch.abs((opt - ref) * color_ref[..., 3:])) * lambda_chroma


def material_smoothness_grad(kd_grad, ks_grad, nrm_grad, lambda_kd=0.25, lambda_ks=0.1, lambda_nrm=0.0):
    """regularizer.py:47-52"""
    kd_luma = (kd_grad[..., 0] + kd_grad[..., 1] + kd_grad[..., 2]) / 3
    loss = torch.mean(kd_luma * kd_grad[..., -1]) * lambda_kd
    loss = loss + torch.mean(ks_grad[..., :-1] * ks_grad[..., -1:]) * lambda_ks
    loss = loss + torch.mean(nrm_grad[..., :-1] * nrm_grad[..., -1:]) * lambda_nrm
    return loss
