"""SDF / deformation MLPs with the reference's constructors and state_dict keys (geometry/mlp.py:9-45,77-118).

`MLP.forward` runs the fused positional-encoding + MLP kernels (csrc/sdf_mlp.hip, sdf_mlp_bwd.hip) for the reference's working
shape (n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3], d_out=1; train.py:1618-1621).  `MLP.forward_reference` is the layer-by-layer
library-GEMM path: it is what the eikonal term's double backward uses (hmsdf.py:856-876; the fused op is first-order), and what
other shapes / `use_float16` fall back to -- on the GPU, never on the CPU."""
import torch
import torch.nn as nn

from d3h import sdf_mlp as _S
from .embedding import Embedding


def _build(emb_dim, extra_in, n_hidden, d_hidden, d_out, skip_in):
    layers = [nn.Linear(emb_dim + extra_in, d_hidden), nn.Softplus(beta=100)]
    skip_count, count = [], 2
    for i in range(n_hidden):
        if i in skip_in:
            layers.append(nn.Linear(d_hidden + emb_dim, d_hidden))
            skip_count.append(count)
        else:
            layers.append(nn.Linear(d_hidden, d_hidden))
        layers.append(nn.Softplus(beta=100))
        count += 2
    layers.append(nn.Linear(d_hidden, d_out))
    return nn.ModuleList(layers), skip_count


class MLP(nn.Module):
    def __init__(self, n_freq=6, d_hidden=128, d_out=1, n_hidden=3, skip_in=[], use_float16=False):
        super().__init__()
        self.emb = Embedding(3, n_freq)
        self.skip_in = skip_in
        self.net, self.skip_count = _build(self.emb.out_channels, 0, n_hidden, d_hidden, d_out, skip_in)
        self.use_float16 = use_float16
        self.fused = (n_freq == 6 and d_hidden == 256 and d_out == 1 and n_hidden == 6 and list(skip_in) == [3] and not use_float16)

    def _params(self):
        # (cached: walking the ModuleList costs ~40 us per call and a training iteration asks four times; nn.Module keeps Parameter OBJECTS
        # across load_state_dict / .to() / optimiser steps -- a replaced first weight or last bias rebuilds the list)
        c = self.__dict__.get('_plist')
        if c is None or c[0] is not self.net[0].weight or c[-1] is not self.net[-1].bias:
            c = [p for i in range(0, len(self.net), 2) for p in (self.net[i].weight, self.net[i].bias)]
            self.__dict__['_plist'] = c
        return c

    def forward_reference(self, x):
        emb = self.emb(x)
        h = emb
        with torch.autocast('cuda', dtype=torch.float16, enabled=self.use_float16):
            for i, m in enumerate(self.net):
                h = m(torch.cat([h, emb], dim=-1)) if i in self.skip_count else m(h)
        return h

    def pack(self):
        """both packed copies of the current weights (d3h.sdf_mlp.PackedWeights) to share between a sweep, its backward and the eikonal
        term of the same iteration; None on the library path"""
        return _S.PackedWeights(self._params()) if self.fused else None

    def forward(self, x, deform=None, disp=0.0, pack=None, rows=None):
        """x [N,3] -> [N,1].  (deform, disp): optional fused `x + disp * deform` (hmsdf.py:433).  rows = (lo, hi): points lo..hi-1 only
        (one rank's shard of a sweep: d3h.dist_ops), gradients in the full-size buffers."""
        if not self.fused:
            v = x if deform is None else x + disp * deform
            return self.forward_reference(v if rows is None else v[rows[0]:rows[1]])
        return _S.sdf_query(x, self._params(), deform=deform, disp=disp, pack=pack, rows=rows)


    def input_gradient(self, x):
        """d(sdf)/d(x) at constant points, differentiable w.r.t. the weights: the graph autograd.grad(..., create_graph=True)
        builds in the reference's eikonal term (hmsdf.py:856-876), as one fused second-order op when the shape allows."""
        if self.fused:
            return _S.sdf_gradient(x.detach(), self._params())
        v = x.detach().requires_grad_(True)
        return torch.autograd.grad(self.forward_reference(v).sum(), v, create_graph=True)[0]


    def eikonal_begin(self, x, pack=None, max_cus=0):
        """queues the forward sweep of eikonal_loss only (fused path); pass the result as `begun=`"""
        return _S.eikonal_begin(x.detach(), self._params(), pack=pack, max_cus=max_cus)

    def eikonal_loss(self, x, coeff, pack=None, begun=None, max_cus=0):
        """coeff * mean((|d sdf/d x| - 1)^2) (hmsdf.py:874-876) -- one fused op with eagerly computed parameter gradients when fused.
        max_cus: the sweeps of the term use at most this many CUs (0 = the chip) -- a per-call launch argument of the C ABI"""
        if self.fused:
            return _S.eikonal_loss(x.detach(), self._params(), coeff, pack=pack, begun=begun, max_cus=max_cus)
        g = self.input_gradient(x)
        return coeff * (g.pow(2).sum(dim=-1).sqrt() - 1).pow(2).mean()


class MLP_deform(nn.Module):
    """Pose-conditioned non-rigid offset network (geometry/mlp.py:77-118); seq-stage component.  For the reference's working shape
    (n_freq 8, d_hidden 256, n_hidden 6, skip_in [3], d_out 3) `forward` runs the fused kernels of csrc/deform_mlp*.hip (the SDF
    kernels built with a 51-wide encoding and a 3-output head, the constant pose code folded into the first bias);
    `forward_reference` is the plain library-GEMM formulation (any shape, gradients w.r.t. the points)."""

    def __init__(self, n_freq=6, d_hidden=128, d_out=1, n_hidden=3, skip_in=[], use_float16=False):
        super().__init__()
        self.emb = Embedding(3, n_freq)
        self.skip_in = skip_in
        self.net, self.skip_count = _build(self.emb.out_channels, 136, n_hidden, d_hidden, d_out, skip_in)
        self.use_float16 = use_float16
        from d3h import deform_mlp as _DM
        self.fused = n_freq == 8 and list(skip_in) == [3] and _DM.supported(self)

    def forward_reference(self, x, code):
        emb = self.emb(x)
        h = torch.cat([code.expand(emb.shape[0], emb.shape[1], -1), emb], dim=-1)
        for i, m in enumerate(self.net):
            h = m(torch.cat([h, emb], dim=-1)) if i in self.skip_count else m(h)
        return h

    def forward(self, x, code):
        if self.fused and not x.requires_grad and code.numel() == 136:
            from d3h import deform_mlp as _DM
            return _DM.offsets(x, code, [p for i in range(0, len(self.net), 2) for p in (self.net[i].weight, self.net[i].bias)])
        return self.forward_reference(x, code)
