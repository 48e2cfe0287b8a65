"""Oracle for mesh normals, shading normal, image loss, SSIM and the SDF edge regulariser (torch CPU; TEST INFRASTRUCTURE).

Restates render/mesh.py:418-446 (auto_normals), render/render.py:261-264 (face normals), render/renderutils/bsdf.py:25-51
(prepare_shading_normal python twin of normal.cu), render/renderutils/c_src/loss.cu:31-131 (the LIVE image-loss path; note the
python validation twin loss.py:17-19 multiplies by exposure=5 before sRGB while the CUDA kernel does not -- the kernel is what
train.py runs, ops.py:496), ssim_loss.py:22-63 and geometry/hmsdf.py:162-170.  Pinned by tests/golden/imgops.npz.
"""
import math

import torch
import torch.nn.functional as F


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(_dot(x, x), min=eps))


def auto_normals(v, f):
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)
    vn = torch.zeros_like(v)
    for c in range(3):
        vn = vn.index_add(0, f[:, c], fn)
    vn = torch.where(_dot(vn, vn) > 1e-20, vn, torch.tensor([0.0, 0.0, 1.0]))
    return safe_normalize(vn)


def face_normals(v, f):
    return safe_normalize(torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=-1))


def prepare_shading_normal(pos, view_pos, pert, snrm, stng, gnrm, two_sided=True, opengl=True):
    N = lambda x: F.normalize(x, dim=-1)
    sn, st, vv = N(snrm), N(stng), N(view_pos - pos)
    bt = N(torch.cross(st.expand_as(sn) if st.shape != sn.shape else st, sn, dim=-1))
    sgn = -1.0 if opengl else 1.0
    sh = N(st * pert[..., 0:1] + sgn * bt * pert[..., 1:2] + sn * torch.clamp(pert[..., 2:3], min=0.0))
    g = gnrm
    if two_sided:
        front = _dot(g, vv) > 0
        sh = torch.where(front, sh, -sh)
        g = torch.where(front, g, -g)
    t = torch.clamp(_dot(vv, sh) / 0.1, min=0, max=1)
    return torch.lerp(g, sh, t)


def _srgb(x):
    return torch.where(x > 0.0031308, torch.pow(torch.clamp(x, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * torch.clamp(x, min=0.0))


def image_loss(img, target, loss='l1', tonemapper='none'):
    """loss.cu:95-131 then ops.py:497: sum(channel-mean loss) / (B*H*W)"""
    a, t = torch.clamp(img, 0, 65535), torch.clamp(target, 0, 65535)
    if tonemapper == 'log_srgb':
        a, t = _srgb(torch.log(a + 1)), _srgb(torch.log(t + 1))
    if loss == 'mse':
        l = (a - t) ** 2
    elif loss == 'relmse':
        l = (a - t) ** 2 / (a * a + t * t + 0.1)
    elif loss == 'smape':
        l = (a - t).abs() / (a + t + 0.01)
    else:
        l = (a - t).abs()
    return l.mean()


def ssim(a, b):
    g = torch.tensor([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum())[:, None]
    C = a.shape[-3]
    win = (g @ g.t()).to(a.dtype)[None, None].expand(C, 1, 11, 11).contiguous().to(a.device)
    conv = lambda x: F.conv2d(x, win, padding=5, groups=C)
    mu1, mu2 = conv(a), conv(b)
    s11, s22, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))).mean()


def sdf_reg_loss(sdf, edges):
    s = sdf.reshape(-1)[edges.reshape(-1)].reshape(-1, 2)
    m = torch.sign(s[:, 0]) != torch.sign(s[:, 1])
    s = s[m]
    return F.binary_cross_entropy_with_logits(s[:, 0], (s[:, 1] > 0).to(s.dtype)) + \
        F.binary_cross_entropy_with_logits(s[:, 1], (s[:, 0] > 0).to(s.dtype))
