"""Scalar loss bookkeeping of tick_* in two launches.

The reference combines its per-pixel means, regularisers and schedule weights with a few dozen scalar tensor operations
(geometry/hmsdf.py:835-915; train.py:718).  On the GPU every one of them is a kernel launch plus an autograd node whose backward
launches more, and that stretch of the iteration is host-bound (the GPU idles between 2 us kernels).  All of those combinations are
affine in the raw terms, so they are evaluated as ONE matrix-vector product  y = M x + c  whose rows are the named losses; the
backward is  dx = M^T dy.  M and c are constants of the run (built once on the device)."""
import torch

_ZERO = {}


def _zero(dev):
    z = _ZERO.get(dev)
    if z is None:
        z = _ZERO[dev] = torch.zeros((), dtype=torch.float32, device=dev)
    return z


class _AffineHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M, c):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(M)
        return tuple(torch.addmv(c, M, x).unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        M, = ctx.saved_tensors
        z = _zero(M.device)
        g = torch.stack([z if v is None else v.reshape(()).float() for v in gs])
        return torch.mv(M.t(), g), None, None


class AffineHead:
    """rows: {name: ({column: coefficient}, constant)} over `ncols` raw terms"""

    def __init__(self, rows, ncols, device):
        self.names = list(rows)
        M = torch.zeros(len(self.names), ncols, dtype=torch.float32)
        c = torch.zeros(len(self.names), dtype=torch.float32)
        for i, n in enumerate(self.names):
            coef, const = rows[n]
            for j, v in coef.items():
                M[i, j] = v
            c[i] = const
        self.M, self.c = M.to(device), c.to(device)

    def __call__(self, x):
        return dict(zip(self.names, _AffineHeadFn.apply(x, self.M, self.c)))
