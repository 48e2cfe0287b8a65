"""Shim for kaolin.ops.mesh.sample_points (call sites geometry/hmsdf.py:714,750): area-weighted face pick + uniform barycentric
sample, p = (1 - sqrt(u)) a + sqrt(u) (1 - v) b + sqrt(u) v c.  kaolin is un-vendored and stochastic: distributional parity only."""
import torch


def sample_points(vertices, faces, num_samples, areas=None, face_features=None):
    v = vertices[0]
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    if areas is None:
        areas = 0.5 * torch.linalg.norm(torch.cross(b - a, c - a, dim=-1), dim=-1)
    pick = torch.multinomial(areas.clamp(min=1e-20), num_samples, replacement=True)
    u = torch.sqrt(torch.rand(num_samples, 1, device=v.device))
    w = torch.rand(num_samples, 1, device=v.device)
    pts = (1 - u) * a[pick] + u * (1 - w) * b[pick] + u * w * c[pick]
    return pts[None], pick[None]
