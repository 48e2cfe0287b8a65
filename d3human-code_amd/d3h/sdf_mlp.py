"""Host side of the fused SDF-query kernels (csrc/sdf_mlp.hip): weight packing + forward (+ backward).

The network shape is the reference's fixed working point (train.py:1618-1621): MLP(n_freq=6, d_hidden=256,
n_hidden=6, skip_in=[3]) from geometry/mlp.py:10-32; any other shape raises (no silent fallback).
"""
import torch

from . import _lib as L

HIDDEN_KEYS = (2, 4, 6, 10, 12)


def check_shape(sd, prefix='net.'):
    want = {0: (256, 39), 2: (256, 256), 4: (256, 256), 6: (256, 256), 8: (256, 295), 10: (256, 256), 12: (256, 256),
            14: (1, 256)}
    for i, shp in want.items():
        w = sd.get(f'{prefix}{i}.weight')
        if w is None or tuple(w.shape) != shp:
            raise RuntimeError(f'd3h.sdf_mlp: unsupported MLP shape at {prefix}{i} '
                               f'({None if w is None else tuple(w.shape)}; kernel is built for n_freq=6,d_hidden=256,'
                               f'n_hidden=6,skip_in=[3])')


def pack_weights(sd, prefix='net.', out=None):
    """state_dict-like {net.i.weight, net.i.bias} -> packed fragment-order buffer (see csrc/sdf_mlp_layout.h)."""
    check_shape(sd, prefix)
    lib = L.lib()
    g = lambda k: sd[prefix + k].detach().contiguous().float()
    wh = torch.stack([g(f'{i}.weight') for i in HIDDEN_KEYS]).contiguous()
    bh = torch.stack([g(f'{i}.bias') for i in HIDDEN_KEYS]).contiguous()
    keep = [g('0.weight'), g('0.bias'), wh, bh, g('8.weight'), g('8.bias'), g('14.weight'), g('14.bias')]
    if out is None:
        out = torch.empty(lib.d3h_sdf_mlp_wpack_floats(), dtype=torch.float32, device=keep[0].device)
    L.check(lib.d3h_sdf_mlp_pack(*[L.ptr(t) for t in keep], L.ptr(out), L.stream()), 'sdf_mlp_pack')
    return out


def forward(x, wpack, deform=None, disp=0.0, save=False, want_xdef=False):
    """sdf[n] (and optionally the saved activations / deformed points) for points x[n,3]."""
    lib = L.lib()
    x = x.contiguous().float()
    n = x.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=x.device)
    act = torch.empty(lib.d3h_sdf_mlp_act_floats(n), dtype=torch.float32, device=x.device) if save else None
    xdef = torch.empty(n, 3, dtype=torch.float32, device=x.device) if want_xdef else None
    d = deform.contiguous().float() if deform is not None else None
    L.check(lib.d3h_sdf_mlp_fwd(L.ptr(x), L.ptr(d), L.f32(disp), L.ptr(wpack), L.ptr(sdf), L.ptr(xdef), L.ptr(act), L.i64(n),
                                L.stream()), 'sdf_mlp_fwd')
    if save or want_xdef:
        return sdf, act, xdef
    return sdf
