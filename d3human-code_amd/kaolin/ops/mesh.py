"""Shim for kaolin.ops.mesh.sample_points (call sites geometry/hmsdf.py:714,750): area-weighted face pick + uniform barycentric
sample, p = (1 - sqrt(u)) a + sqrt(u) (1 - v) b + sqrt(u) v c.  kaolin is un-vendored and stochastic: distributional parity only.

When no gradient is being recorded (the hot path samples under no_grad: the eikonal term treats the points as constants) the face
areas and the barycentric map are one kernel each (csrc/image_ops.hip); torch supplies the random numbers.  With autograd on, the
same formulas run as differentiable torch ops."""
import os

import torch

FUSED_SAMPLER = os.environ.get('D3H_FUSED_SAMPLER', '1') != '0'      # '0': the torch.multinomial form (A/B)


def _pick_weights(areas):
    """multinomial weights: the areas themselves -- zero-area rows (degenerate faces, the zero padding of a face list at its allocation
    bound) have probability exactly 0, as with kaolin's Categorical(areas) -- except when EVERY area is zero (no mesh): then uniform, so that
    multinomial stays defined; the caller discards such samples (geometry/hmsdf.py:_extract).  No host synchronisation."""
    return areas + (areas.sum() <= 0).to(areas.dtype) * 1e-20


def sample_points(vertices, faces, num_samples, areas=None, face_features=None, _rnd=None):
    v = vertices[0]
    if not (torch.is_grad_enabled() and (v.requires_grad or (areas is not None and areas.requires_grad))):
        from d3h import _lib as L
        vc = v.detach().contiguous().float()
        fc = faces.contiguous()
        if fc.dtype != torch.int64:
            fc = fc.long()
        nf = fc.shape[0]
        if areas is None and nf > 0 and FUSED_SAMPLER:
            # the whole sampler as one call (two kernels: area prefix sums in one workgroup, inverse-CDF pick + barycentric map per sample) on
            # ONE torch.rand -- 3 launches where the multinomial form below takes 14, on the host-bound stretch right after the marching-tets
            # read-back (tools/dbg/gpu_host_sync_timing.py).  Same distribution (area-weighted with replacement, zero-area rows never picked);
            # not the same draws as torch.multinomial on the same seed
            # (_rnd: the [num_samples, 3] uniform numbers, drawn by a caller that may have to repeat the call on the same draws)
            rnd = _rnd if _rnd is not None else torch.rand(num_samples, 3, device=v.device)
            pts = torch.empty(num_samples, 3, dtype=torch.float32, device=v.device)
            pick = torch.empty(num_samples, dtype=torch.int64, device=v.device)
            cdf = torch.empty(nf, dtype=torch.float32, device=v.device)
            L.check(L.lib().d3h_sample_surface(L.ptr(vc), L.ptr(fc), L.i32(nf), L.ptr(rnd), L.i32(num_samples), L.ptr(cdf), L.ptr(pts), L.ptr(pick),
                                               L.stream()), 'sample_surface')
            return pts[None], pick[None]
        if areas is None:
            areas = torch.empty(nf, dtype=torch.float32, device=v.device)
            L.check(L.lib().d3h_face_areas(L.ptr(vc), L.ptr(fc), L.i32(nf), L.ptr(areas), L.stream()), 'face_areas')
        pick = torch.multinomial(_pick_weights(areas), num_samples, replacement=True)
        uw = torch.rand(num_samples, 2, device=v.device)
        pts = torch.empty(num_samples, 3, dtype=torch.float32, device=v.device)
        L.check(L.lib().d3h_sample_faces(L.ptr(vc), L.ptr(fc), L.ptr(pick), L.ptr(uw), L.i32(num_samples), L.ptr(pts), L.stream()), 'sample_faces')
        return pts[None], pick[None]
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    if areas is None:
        areas = 0.5 * torch.linalg.norm(torch.cross(b - a, c - a, dim=-1), dim=-1)
    pick = torch.multinomial(_pick_weights(areas), num_samples, replacement=True)
    u = torch.sqrt(torch.rand(num_samples, 1, device=v.device))
    w = torch.rand(num_samples, 1, device=v.device)
    pts = (1 - u) * a[pick] + u * (1 - w) * b[pick] + u * w * c[pick]
    return pts[None], pick[None]
