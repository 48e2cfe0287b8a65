"""Collectives of the frame-parallel step that sit INSIDE the autograd graph (RCCL through torch.distributed, backend "nccl" on the
GPUs, gloo in the CPU tests).

`gather_shards`: the SDF sweep over the tet grid is identical on every rank (shared canonical geometry), so with W ranks each one
evaluates only N_v / W grid vertices and the values are all-gathered (1 MB at tet-res 128).  In the backward every rank holds
d(loss_r)/d(sdf) of ITS frames for all vertices; the all-reduce(sum) hands each rank the summed gradient on its own shard, which it
back-propagates through its part of the sweep.  The later bucket all-reduce of the parameter gradients (Scene.allreduce_grads, sum / W)
then yields exactly the mean-over-ranks gradient of the unsharded computation: sum over shards of (sum over ranks) / W."""
import torch
import torch.distributed as dist


def _is_nccl():
    return dist.get_backend() == 'nccl'


class _GatherShardsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, n_total, shard, rank, world):
        pad = local
        if local.shape[0] != shard:                      # the last shard may be short: the collective wants equal sizes
            pad = local.new_zeros((shard,) + tuple(local.shape[1:]))
            pad[:local.shape[0]] = local
        out = torch.empty((world * shard,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad.contiguous())          # one flat output: no list of parts, no cat
        ctx.meta = (local.shape[0], shard, rank, world, n_total)
        return out[:n_total]

    @staticmethod
    def backward(ctx, g):
        n_local, shard, rank, world, n_total = ctx.meta
        g = g.contiguous()
        if _is_nccl():
            # every rank needs the summed gradient of ITS shard only: reduce-scatter moves 1/W of what an all-reduce would and reads the
            # engine's gradient buffer without modifying it (an in-place all-reduce needed a private 1 MB copy first)
            if n_total != world * shard:
                gp = g.new_zeros((world * shard,) + tuple(g.shape[1:]))
                gp[:n_total] = g
                g = gp
            mine = torch.empty((shard,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(mine, g, op=dist.ReduceOp.SUM)
            return mine[:n_local], None, None, None, None
        g = g.clone()                                    # gloo (CPU tests) has no reduce-scatter
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[rank * shard: rank * shard + n_local], None, None, None, None


def shard_range(n_total, rank, world, align=128):
    """[lo, hi) of rank's slice of n_total items and the (aligned) shard size; align = the point-tile size of the SDF kernels"""
    shard = (((n_total + world - 1) // world) + align - 1) // align * align
    lo = min(rank * shard, n_total)
    return lo, min(lo + shard, n_total), shard


def gather_shards(local, n_total, shard, rank, world):
    return _GatherShardsFn.apply(local, n_total, shard, rank, world)
