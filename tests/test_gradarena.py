"""d3h.gradarena.GradArena on the CPU (pure torch, no kernels): the contract the frame-parallel step relies on -- leaf gradients produced
inside ONE flat buffer that the all-reduce runs on in place (d3h/scene.py:allreduce_grads)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))


class _Scale(torch.autograd.Function):
    """y = a * p with the gradient of the LEAF p delivered through the arena (what the sweep / texture backward do)"""

    @staticmethod
    def forward(ctx, p, a, rows):
        ctx.a, ctx.leaf, ctx.rows = a, p, rows
        return (p if rows is None else p[rows[0]:rows[1]]) * a

    @staticmethod
    def backward(ctx, g):
        from d3h import gradarena as GA
        return GA.deliver(ctx.leaf, g, ctx.a, rows=ctx.rows), None, None


def test_arena_slices_become_the_grads_and_the_bucket_is_complete():
    from d3h import gradarena as GA
    p = torch.nn.Parameter(torch.arange(12.0).reshape(4, 3))
    q = torch.nn.Parameter(torch.ones(5))
    r = torch.nn.Parameter(torch.zeros(2, 2))                 # never receives a gradient: contributes zeros
    outside = torch.nn.Parameter(torch.ones(3))               # not a member: plain autograd
    ar = GA.GradArena([p, q, r])
    assert ar.numel == 12 + 5 + 4
    ar.flat.fill_(7.0)                                        # stale content of the previous step
    ar.begin()
    assert GA.ACTIVE is ar and not ar.flat.any()
    # two contributions to p (one covering rows 1..3 only), one to q through plain autograd (no producer asks for its slice)
    y = _Scale.apply(p, 2.0, None).sum() + _Scale.apply(p, 3.0, (1, 3)).sum() + (q * q).sum() + _Scale.apply(outside, 5.0, None).sum()
    y.backward()
    vp = ar.views[0]
    assert p.grad.data_ptr() == vp.data_ptr()                 # AccumulateGrad kept the arena view: .grad IS the bucket slice
    assert torch.equal(p.grad, torch.tensor([[2.0] * 3, [5.0] * 3, [5.0] * 3, [2.0] * 3]))
    assert torch.equal(outside.grad, torch.full((3,), 5.0)) and outside.grad.data_ptr() != ar.flat.data_ptr()
    assert q.grad.data_ptr() != ar.views[1].data_ptr()        # produced elsewhere ...
    flat = ar.collect()
    assert GA.ACTIVE is None
    assert q.grad.data_ptr() == ar.views[1].data_ptr() and torch.equal(q.grad, torch.full((5,), 2.0))      # ... reconciled by collect()
    assert r.grad.data_ptr() == ar.views[2].data_ptr() and not r.grad.any()
    assert torch.equal(flat, torch.cat([p.grad.reshape(-1), q.grad, r.grad.reshape(-1)]))
    flat.mul_(0.5)                                            # "the all-reduce": in place, and the .grad tensors see it
    assert torch.equal(p.grad[0], torch.full((3,), 1.0)) and torch.equal(q.grad, torch.full((5,), 1.0))


def test_block_of_adjacent_members_and_inactive_arena():
    from d3h import gradarena as GA
    a, b, c = (torch.nn.Parameter(torch.zeros(n)) for n in (3, 4, 2))
    ar = GA.GradArena([a, b, c])
    assert GA.slot_for(a) is None and GA.block_for([a, b]) is None          # no step in flight: producers fall back to fresh buffers
    ar.begin()
    blk = GA.block_for([a, b])
    assert blk is not None and blk.numel() == 7 and blk.data_ptr() == ar.flat.data_ptr()
    assert GA.block_for([a, b]) is None                                     # handed out once
    assert GA.accum_block_for([a, b]).data_ptr() == blk.data_ptr()          # a later contribution accumulates onto it
    assert GA.block_for([c, a]) is None                                     # not adjacent in this order
    s1, s2 = GA.slot_for(c), GA.slot_for(c)
    assert s1 is not None and s2 is None and GA.accum_for(c).data_ptr() == s1.data_ptr()
    assert s1 is not ar.views[2]                                            # a FRESH tensor object (AccumulateGrad only keeps an unshared one)
    # a handed slice whose result never reached .grad (a partial backward) is no gradient
    s1.fill_(9.0)
    ar.collect()
    assert not c.grad.any() and not a.grad.any()
    # without begin() (a caller that set .grad by hand): members without a gradient read zero, whatever the buffer held
    ar.flat.fill_(3.0)
    a.grad, b.grad, c.grad = None, torch.full((4,), 2.0), None
    flat = ar.collect()
    assert torch.equal(flat, torch.tensor([0.0] * 3 + [2.0] * 4 + [0.0] * 2))


def test_plain_contribution_between_slot_and_accum_is_never_silently_dropped():
    """slot() hands the slice to autograd; a PLAIN autograd contribution to the same leaf makes the engine sum out of place
    (.grad becomes a new tensor); a later accum() then lands on the slice only.  collect() used to overwrite the slice with .grad and lose
    the accumulated part: it must raise instead (the total cannot be reconstructed).  Without an accum() the foreign sum is reconciled."""
    import pytest
    from d3h import gradarena as GA
    p = torch.nn.Parameter(torch.ones(4))
    ar = GA.GradArena([p])
    ar.begin()
    # arena-aware first producer, a plain autograd consumer of the same leaf, then a second arena-aware producer (accum path of deliver)
    y = _Scale.apply(p, 2.0, None).sum() + (p * p).sum() + _Scale.apply(p, 3.0, None).sum()
    y.backward()
    assert 0 in ar.accumulated
    if p.grad.data_ptr() != ar.views[0].data_ptr():           # the engine summed out of place: the ambiguous case
        with pytest.raises(RuntimeError, match='cannot be reconstructed'):
            ar.collect()
    else:                                                      # (an engine that adds in place keeps everything in the slice)
        ar.collect()
        assert torch.equal(p.grad, torch.full((4,), 7.0))
    GA.ACTIVE = None
    # the same without a second arena-aware producer: nothing was added in place, .grad is complete, collect() copies it into the slice
    p.grad = None
    ar.begin()
    (_Scale.apply(p, 2.0, None).sum() + (p * p).sum()).backward()
    ar.collect()
    assert p.grad.data_ptr() == ar.views[0].data_ptr() and torch.equal(p.grad, torch.full((4,), 4.0))
