"""One flat fp32 arena for the gradients of the shared parameters of a frame-parallel step.

The step's only collective is ONE all-reduce over this buffer (Scene.allreduce_grads).  The kernels that produce leaf gradients write
into it directly: while an arena is active (`begin()` ... `collect()`), `slot_for(param)` hands the backward of the SDF sweep (deform),
of the texture MLP (grid table, weights) and of the flat SDF parameter vector the arena slice of that parameter instead of a fresh
buffer, autograd's AccumulateGrad keeps the returned view as `.grad` (it takes a gradient over when the leaf has none), and the
all-reduce runs in place: no flatten `cat` before the collective, no copy back after it, and the per-buffer zero fills of those
gradients become the arena's single fill.  A gradient that arrives some other way (a second contribution that autograd adds, a
foreign producer, a parameter without a gradient on this rank) is reconciled by `collect()`: the arena is always the complete,
zero-padded bucket with the same layout on every rank.
"""
import torch

ACTIVE = None           # the arena of the step in flight (one per process: one Python thread per GPU)


class GradArena:
    def __init__(self, params):
        """params: the bucket members in layout order (every shared parameter that requires a gradient)"""
        self.params = list(params)
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += p.numel()
        self.numel = o
        dev = self.params[0].device if self.params else 'cpu'
        self.flat = torch.zeros(o, dtype=torch.float32, device=dev)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.views = [self.flat[a:a + p.numel()].view(p.shape) for a, p in zip(self.offsets, self.params)]
        self.handed = set()
        self.accumulated = set()          # members whose slice received an in-place accum() / accum_block() contribution this step
        self.began = False

    def begin(self):
        """start of a step (after zero_grad): one fill for every gradient of the bucket"""
        global ACTIVE
        self.flat.zero_()
        self.handed.clear()
        self.accumulated.clear()
        self.began = True
        ACTIVE = self

    def slot(self, p):
        """the arena slice of parameter `p` for a producer that WRITES the whole gradient (or accumulates onto zeros); None when `p` is not a
        member, already holds a gradient (autograd would add the producer's result to it), or its slice was handed out before"""
        i = self.index.get(id(p))
        if i is None or p.grad is not None or i in self.handed or p.dtype != torch.float32:
            return None
        self.handed.add(i)
        a = self.offsets[i]
        return self.flat[a:a + p.numel()].view(p.shape)          # a FRESH tensor object: AccumulateGrad only keeps a gradient nobody else references

    def block(self, params):
        """one flat view over `params` if they are adjacent in the arena in this order (the 16 tensors of the SDF network in the order
        the gradient kernels write them), else None"""
        idx = [self.index.get(id(p)) for p in params]
        if not idx or idx[0] is None or any(p.grad is not None for p in params):
            return None
        if any(j != idx[0] + k for k, j in enumerate(idx)) or any(j in self.handed for j in idx):
            return None
        self.handed.update(idx)
        a = self.offsets[idx[0]]
        return self.flat[a:a + sum(p.numel() for p in params)]

    def accum(self, p):
        """a LATER contribution of the same backward pass to a parameter whose slice was handed out by slot(): the producer accumulates
        onto the slice (atomics onto what the first producer wrote) and returns no gradient for it -- nothing is left for autograd to add"""
        i = self.index.get(id(p))
        if i is None or i not in self.handed or p.grad is not None:
            return None
        self.accumulated.add(i)
        a = self.offsets[i]
        return self.flat[a:a + p.numel()].view(p.shape)

    def accum_block(self, params):
        idx = [self.index.get(id(p)) for p in params]
        if not idx or idx[0] is None or any(j != idx[0] + k for k, j in enumerate(idx)) or any(j not in self.handed for j in idx) \
                or any(p.grad is not None for p in params):
            return None
        self.accumulated.update(idx)
        a = self.offsets[idx[0]]
        return self.flat[a:a + sum(p.numel() for p in params)]

    def collect(self):
        """end of the backward: make every member's .grad its arena slice (copying a gradient that was produced elsewhere, leaving zeros for
        a member without one); returns the flat buffer"""
        global ACTIVE
        ACTIVE = None
        fresh, self.began = self.began, False          # no begin() for this step (a caller that set .grad by hand): nothing is pre-zeroed
        accumulated, self.accumulated = (self.accumulated if fresh else set()), set()
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                if not fresh or self.index[id(p)] in self.handed:
                    v.zero_()                   # (handed to a producer whose result never reached .grad -- a partial backward -- is no gradient)
                p.grad = v
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape:
                if self.index[id(p)] in accumulated:
                    # The slice was handed to autograd by slot() / block(), a PLAIN autograd contribution to the same leaf arrived before
                    # AccumulateGrad ran (the engine then sums out of place: .grad is a new tensor), and a producer also added onto the slice
                    # in place through accum(): whether that addition is inside .grad depends on when the engine formed its sum -- the two
                    # cannot be told apart afterwards.  Never silently drop a contribution: the producer set of an arena member must be
                    # closed (arena-aware producers only), or the foreign producer must come first / go through deliver().
                    raise RuntimeError('gradarena: parameter #%d received an in-place accum() contribution AND a gradient formed outside the '
                                       'arena in the same step; its total cannot be reconstructed' % self.index[id(p)])
                v.copy_(g)
                p.grad = v
        return self.flat


def slot_for(p):
    return ACTIVE.slot(p) if (ACTIVE is not None and p is not None) else None


def block_for(params):
    return ACTIVE.block(params) if ACTIVE is not None else None


def accum_for(p):
    return ACTIVE.accum(p) if (ACTIVE is not None and p is not None) else None


def accum_block_for(params):
    return ACTIVE.accum_block(params) if ACTIVE is not None else None


def deliver(p, value, alpha=1.0, rows=None):
    """`alpha * value` as (part of) the gradient of leaf `p`, for a backward node: the first contribution of a pass is written into p's arena
    slice and that slice is returned (AccumulateGrad keeps it), a later one is added onto the slice in place and None is returned, and
    without an active arena (or for a non-member) the plain product comes back.  rows = (lo, hi): `value` covers p[lo:hi] only."""
    slot = slot_for(p)
    if slot is not None:
        torch.mul(value, alpha, out=slot if rows is None else slot[rows[0]:rows[1]])
        return slot
    acc = accum_for(p)
    if acc is not None:
        (acc if rows is None else acc[rows[0]:rows[1]]).add_(value, alpha=alpha)
        return None
    if rows is None:
        return value * alpha
    full = torch.zeros_like(p, dtype=value.dtype)
    torch.mul(value, alpha, out=full[rows[0]:rows[1]])
    return full


class _Displace(torch.autograd.Function):
    """verts + disp * deform (hmsdf.py:433; two roundings, as the reference's expression) whose gradient w.r.t. the leaf `deform` goes
    through `deliver`: in a frame-parallel step it meets the SDF sweep's contribution inside the all-reduce arena instead of in a sum
    the autograd engine forms outside it"""

    @staticmethod
    def forward(ctx, verts, deform, disp):
        ctx.disp = float(disp)
        ctx.leaf = deform if deform.is_leaf else None
        return verts + disp * deform

    @staticmethod
    def backward(ctx, g):
        leaf, ctx.leaf = ctx.leaf, None
        gv = g if ctx.needs_input_grad[0] else None
        gd = deliver(leaf, g, ctx.disp) if ctx.needs_input_grad[1] else None
        return gv, gd, None


def displace(verts, deform, disp):
    return _Displace.apply(verts, deform, disp)
