"""Oracle for the multiresolution grid encoding + texture MLP (torch CPU; TEST INFRASTRUCTURE).

PARITY UNPINNED for the encoding: tiny-cuda-nn (README.md:30, un-vendored, unpinned; call site render/mlptexture.py:79,98) is not
in the reference tree; this restates its published HashGrid algorithm for the dense-level configuration of mlptexture.py:62-75.
The MLP / bbox / sigmoid part restates render/mlptexture.py:18-41,91-107.
"""
import math

import torch

PER_LEVEL_SCALE = math.exp(math.log(4096 / 16) / 15)


def grid_layout(per_level_scale=PER_LEVEL_SCALE, base=16, n_levels=5):
    import numpy as np
    out, off = [], 0
    for l in range(n_levels):
        scale = float(np.float32(np.exp2(np.float32(l) * np.log2(np.float32(per_level_scale)))) * np.float32(base) - np.float32(1))
        res = int(math.ceil(scale)) + 1
        size = (res ** 3 + 7) // 8 * 8
        out.append((scale, res, off, size))
        off += size
    return out, off


def grid_encode(x, table):
    """x [N,3] in [0,1] -> [N,10]; table: flat [n_entries*2]"""
    lay, _ = grid_layout()
    tab = table.view(-1, 2)
    feats = []
    for scale, res, off, size in lay:
        p = x * scale + 0.5
        fl = torch.floor(p)
        fr = p - fl
        pg = fl.long()
        acc = 0
        for c in range(8):
            w = 1.0
            idx = 0
            stride = 1
            for d in range(3):
                bit = (c >> d) & 1
                w = w * (fr[:, d] if bit else (1 - fr[:, d]))
                idx = idx + (pg[:, d] + bit) * stride
                stride *= res
            idx = torch.where(idx >= size, idx - size, idx)
            acc = acc + w[:, None] * tab[off + idx]
        feats.append(acc)
    return torch.cat(feats, -1)


def texture_mlp(x, table, w1, w2, w3, bbox, omin, omax, in_grad_scale=128.0):
    b0, b1 = torch.tensor(bbox[:3]), torch.tensor(bbox[3:])
    xn = torch.clamp((x.reshape(-1, 3) - b0) / (b1 - b0), 0, 1)                 # mlptexture.py:94-96
    enc = grid_encode(xn, table)
    if in_grad_scale != 1.0:                                                   # mlptexture.py:31: grad_input * loss_scale
        enc = _ScaleGrad.apply(enc, in_grad_scale)
    h = torch.relu(enc @ w1.t())
    h = torch.relu(h @ w2.t())
    o = h @ w3.t()
    out = torch.sigmoid(o) * (torch.tensor(omax) - torch.tensor(omin)) + torch.tensor(omin)
    return out.reshape(*x.shape[:-1], 6)


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None
