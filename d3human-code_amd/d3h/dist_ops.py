"""Collectives of the frame-parallel step (RCCL through torch.distributed, backend "nccl" on the GPUs, gloo in the CPU tests) and the
work split that goes with them.

What is sharded when W ranks train together (Scene.enable_work_sharding, the default of bench.py for N > 1):
 * `gather_shards`: the SDF sweep over the tet grid is identical on every rank (shared canonical geometry), so each rank evaluates only
   N_v / W grid vertices and the values are all-gathered (1 MB at tet-res 128).  In the backward every rank holds d(loss_r)/d(sdf) of ITS
   frames for all vertices; the reduce-scatter(sum) hands each rank the summed gradient on its own shard, which it back-propagates through
   its part of the sweep.  The later bucket all-reduce of the parameter gradients (Scene.allreduce_grads, mean over ranks) then yields
   exactly the mean-over-ranks gradient of the unsharded computation: sum over shards of (sum over ranks) / W.
 * the eikonal term draws ceil(S / W) surface samples per rank instead of S (hmsdf.py:714: S = 50 000): the mean over ranks of the per-rank
   means is the same estimator over S samples (Scene.enable_work_sharding sets FLAGS.eikonal_samples).
 * NOT sharded, on purpose: marching tets (every rank needs the whole mesh), and the sdf_reg term -- its mean runs over the sign-changing
   edges, so a 1 / W slice of the edge list needs the global count, i.e. one more collective in the forward (>= 25 us of latency) to save
   7/8 of a 27 us kernel.

Virtual-rank mode (`set_virtual(rank, world)`; bench.py --as-rank-of W): ONE process on one GPU runs exactly the work of rank `rank` of a
W-rank job, no process group: the all-gather writes the local shard into a resident full-size buffer whose other shards were filled by
`set_virtual_full` (Scene.refresh_virtual: a full sweep of the current parameters, outside the timed region, with the learning rates at
zero while timing so the foreign shards stay exact), the reduce-scatter returns the local shard of the local gradient, the all-reduce is
the identity.  Every kernel and every byte of glue of a real rank's step runs; only the wire time is missing, and that is modelled
(bench.py: xGMI model of SURVEY section 5)."""
import torch
import torch.distributed as dist

_VIRTUAL = None          # (rank, world) of the virtual-rank mode
_VIRT_FULL = {}          # n_total -> resident [world * shard, ...] buffer holding the other ranks' shards


def set_virtual(rank=None, world=None):
    """enter (rank, world) / leave (no arguments) the virtual-rank mode"""
    global _VIRTUAL
    _VIRTUAL = None if rank is None else (int(rank), int(world))
    _VIRT_FULL.clear()


def virtual():
    return _VIRTUAL


def set_virtual_full(values):
    """the full gathered tensor ([n_total, ...]) of the current parameters: the stand-in for what the other ranks would send"""
    _VIRT_FULL[int(values.shape[0])] = values.detach().clone()


def active():
    """(rank, world) of the real process group or of the virtual-rank mode; (0, 1) otherwise"""
    if _VIRTUAL is not None:
        return _VIRTUAL
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _is_nccl():
    return dist.get_backend() == 'nccl'


def all_reduce_mean(flat, world):
    """in-place mean over the ranks of one flat buffer: ONE collective (ncclAvg on RCCL; gloo, which has no AVG: SUM and a scale)"""
    if _VIRTUAL is not None:
        return flat
    if _is_nccl():
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
    return flat


class _GatherShardsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, n_total, shard, rank, world):
        ctx.meta = (local.shape[0], shard, rank, world, n_total)
        if _VIRTUAL is not None:
            full = _VIRT_FULL.get(int(n_total))
            if full is None:
                raise RuntimeError('dist_ops: virtual-rank mode without set_virtual_full() for this gather')
            out = full.clone()                                   # (what the all-gather writes: one full-size buffer per step)
            out[rank * shard: rank * shard + local.shape[0]] = local
            return out
        pad = local
        if local.shape[0] != shard:                      # the last shard may be short: the collective wants equal sizes
            pad = local.new_zeros((shard,) + tuple(local.shape[1:]))
            pad[:local.shape[0]] = local
        out = torch.empty((world * shard,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad.contiguous())          # one flat output: no list of parts, no cat
        return out[:n_total]

    @staticmethod
    def backward(ctx, g):
        n_local, shard, rank, world, n_total = ctx.meta
        g = g.contiguous()
        if _VIRTUAL is not None:
            return g[rank * shard: rank * shard + n_local].clone(), None, None, None, None
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


def samples_per_rank(total, world):
    """surface samples of the eikonal term each of `world` ranks draws so that together they cover `total` (hmsdf.py:714)"""
    return -(-int(total) // max(1, int(world)))


# ---- xGMI cost model of the step's collectives (used by bench.py's predicted-scaling table) -----------------------------
XGMI_LINK_GBPS = 153.0 / 2          # one direction of one of the 7 point-to-point links of an MI355X (~153 GB/s bidirectional per link)
XGMI_EFFICIENCY = 0.8               # protocol efficiency assumed on top of the link rate
COLLECTIVE_LATENCY_US = 30.0        # launch + synchronisation floor of one small RCCL collective on 8 GPUs (assumed; measured on hardware by the driver's SCALE run)


def model_collective_us(kind, nbytes, world, links=1):
    """microseconds of one collective of `nbytes` (the full buffer) over `world` GPUs of one node.  `links` = 1: a ring bound by ONE link per
    hop (pessimistic on the fully connected 8-GPU node); `links` = world - 1: the direct all-to-all form in which every peer pair uses its own
    link (reduce-scatter + all-gather with 1/W of the buffer per peer)."""
    if world <= 1:
        return 0.0
    bw = XGMI_LINK_GBPS * XGMI_EFFICIENCY * max(1, min(links, world - 1)) * 1e9
    frac = (world - 1) / world
    moved = {'all_reduce': 2 * frac, 'all_gather': frac, 'reduce_scatter': frac}[kind] * nbytes
    return COLLECTIVE_LATENCY_US + moved / bw * 1e6
