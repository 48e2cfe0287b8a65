"""Math / image helpers with the names render/*.py of the reference uses (render/util.py:19-29,61-66,94-110,195-210,237-300).
Display, GLFW and image-file IO of the reference's util.py are outside the hot path and not provided."""
import numpy as np
import torch


def dot(x, y):
    return torch.sum(x * y, -1, keepdim=True)


def reflect(x, n):
    return 2 * dot(x, n) * n - x


def length(x, eps=1e-20):
    return torch.sqrt(torch.clamp(dot(x, x), min=eps))


def safe_normalize(x, eps=1e-20):
    return x / length(x, eps)


def to_hvec(x, w):
    return torch.nn.functional.pad(x, pad=(0, 1), mode='constant', value=w)


_grid_cache = {}


def pixel_grid(width, height, center_x=0.5, center_y=0.5, device=None):
    """util.py:61-66: [H,W,2] of ((x + cx)/W, (y + cy)/H)"""
    device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
    key = (width, height, center_x, center_y, str(device))
    g = _grid_cache.get(key)
    if g is None:
        y, x = torch.meshgrid((torch.arange(0, height, dtype=torch.float32, device=device) + center_y) / height,
                              (torch.arange(0, width, dtype=torch.float32, device=device) + center_x) / width, indexing='ij')
        g = torch.stack((x, y), dim=-1)
        _grid_cache[key] = g
    return g


def _rgb_to_srgb(f):
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(torch.clamp(f, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def rgb_to_srgb(f):
    return torch.cat((_rgb_to_srgb(f[..., 0:3]), f[..., 3:4]), dim=-1) if f.shape[-1] == 4 else _rgb_to_srgb(f)


def _srgb_to_rgb(f):
    return torch.where(f <= 0.04045, f / 12.92, torch.pow((torch.clamp(f, 0.04045) + 0.055) / 1.055, 2.4))


def srgb_to_rgb(f):
    return torch.cat((_srgb_to_rgb(f[..., 0:3]), f[..., 3:4]), dim=-1) if f.shape[-1] == 4 else _srgb_to_rgb(f)


def scale_img_nhwc(x, size, mag='bilinear', min='area'):
    y = x.permute(0, 3, 1, 2)
    if x.shape[1] > size[0] and x.shape[2] > size[1]:
        y = torch.nn.functional.interpolate(y, size, mode=min)
    elif mag in ('bilinear', 'bicubic'):
        y = torch.nn.functional.interpolate(y, size, mode=mag, align_corners=True)
    else:
        y = torch.nn.functional.interpolate(y, size, mode=mag)
    return y.permute(0, 2, 3, 1).contiguous()


def scale_img_hwc(x, size, mag='bilinear', min='area'):
    return scale_img_nhwc(x[None, ...], size, mag, min)[0]


def avg_pool_nhwc(x, size):
    return torch.nn.functional.avg_pool2d(x.permute(0, 3, 1, 2), size).permute(0, 2, 3, 1).contiguous()


def perspective(fovy=0.7854, aspect=1.0, n=0.1, f=1000.0, device=None):
    y = np.tan(fovy / 2)
    return torch.tensor([[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]],
                        dtype=torch.float32, device=device)


def translate(x, y, z, device=None):
    return torch.tensor([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=torch.float32, device=device)
