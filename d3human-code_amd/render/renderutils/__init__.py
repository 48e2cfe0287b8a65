"""render.renderutils with the reference's entry points (render/renderutils/ops.py:197,479,518,541) on the HIP kernels.
The BSDF / cubemap functions of the reference plugin are dead under the hard-wired bsdf='kd' (render/render.py:120) and are
not provided."""
import logging
import os

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


_IMAGE_LOSS_PROBE = None       # a list while geometry.hmsdf probes a loss callable (see loss_spec): every call is recorded in it


def image_loss(img, target, loss='l1', tonemapper='none', use_python=False):
    out = _I.image_loss(img, target, loss=loss, tonemapper=tonemapper)
    if _IMAGE_LOSS_PROBE is not None:
        _IMAGE_LOSS_PROBE.append((img, target, loss, tonemapper, out))
    return out


_SPEC_CACHE = {}
_LOG = logging.getLogger('d3h.loss_spec')


def loss_spec(loss_fn, device):
    """(loss, tonemapper) when `loss_fn(img, ref)` IS `image_loss(img, ref, loss=..., tonemapper=...)` -- what train.py:75-87 (createLoss)
    builds as bare lambdas -- else None.  A callable may declare it (`loss_fn.d3h_spec = ('l1', 'log_srgb')`); otherwise it is called once on
    two one-pixel images with this module's image_loss recording: exactly one call, on exactly those tensors, whose result comes back
    untouched, identifies it.  tick_* then evaluate the term inside their fused per-pixel pass instead of through separate mask products,
    slices and the stand-alone image-loss kernel.  Cached per callable object (the entry holds the callable, so its id cannot be re-used)."""
    spec = getattr(loss_fn, 'd3h_spec', None)
    if spec is not None:
        return tuple(spec)
    if os.environ.get('D3H_LOSS_PROBE', '1') == '0':          # (A/B switch: undeclared callables take the separate-kernel path)
        return None
    # a bound method is a new object on every attribute access: identify it by its function and its instance
    ident = (getattr(loss_fn, '__func__', loss_fn), getattr(loss_fn, '__self__', None))
    key = (id(ident[0]), id(ident[1]), str(device))
    hit = _SPEC_CACHE.get(key)
    if hit is not None and hit[0][0] is ident[0] and hit[0][1] is ident[1]:
        return hit[1]
    global _IMAGE_LOSS_PROBE
    found = None
    try:
        a = torch.full((1, 1, 1, 3), 0.25, dtype=torch.float32, device=device)
        b = torch.full((1, 1, 1, 3), 0.5, dtype=torch.float32, device=device)
        _IMAGE_LOSS_PROBE = calls = []
        with torch.no_grad():
            out = loss_fn(a, b)
        if len(calls) == 1 and calls[0][0] is a and calls[0][1] is b and out is calls[0][4] and calls[0][2] in _I._LOSS and calls[0][3] in _I._TONE:
            found = (calls[0][2], calls[0][3])
    except Exception:
        found = None
    finally:
        _IMAGE_LOSS_PROBE = None
    if len(_SPEC_CACHE) > 16:
        _SPEC_CACHE.clear()
    _SPEC_CACHE[key] = (ident, found)
    # the decision is taken ONCE per callable from a one-pixel call: a stateful or shape-dependent callable could be misjudged, so say what
    # was decided (once, through `logging`: logger "d3h.loss_spec"; D3H_LOSS_PROBE=0 or `loss_fn.d3h_spec = ...` override it)
    name = getattr(ident[0], '__qualname__', None) or repr(ident[0])
    if found is not None:
        _LOG.info("loss callable %s recognised as image_loss(loss=%r, tonemapper=%r): evaluated inside the fused per-pixel pass", name, *found)
    else:
        _LOG.info("loss callable %s is not a bare image_loss call: evaluated through its own kernels on the masked images", name)
    return found


def prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True, use_python=False):
    return _I.prepare_shading_normal(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading, opengl)
