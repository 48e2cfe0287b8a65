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
COLLECTIVE_LATENCY_US = 30.0        # launch + synchronisation floor of one small RCCL collective: an ASSUMPTION until measure_rccl_floor_us() replaces it
MEASURED_FLOOR_US = None            # {'all_gather': us, 'reduce_scatter': us, 'all_reduce': us} of the step's three collectives at their real sizes, once measured


def measure_rccl_floor_us(n_grid, bucket_bytes, device, reps=30):
    """The per-call floor of the step's three collectives AT THEIR REAL SIZES on whatever process group is initialised (the one-rank RCCL group of
    a single-GPU box gives the launch + kernel floor: no wire): all_gather_into_tensor and reduce_scatter_tensor of the 4 n_grid-byte sdf /
    d(sdf) vectors, all_reduce(AVG) of the gradient bucket.  HIP events on the current stream, mean of `reps` after a warm-up.  The result
    replaces the assumed COLLECTIVE_LATENCY_US in model_collective_us (per kind) and is reported by bench.py (`config.rccl_floor_us`)."""
    global MEASURED_FLOOR_US
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError('measure_rccl_floor_us needs an initialised process group')
    W = dist.get_world_size()
    shard = -(-int(n_grid) // W)
    loc, full = torch.zeros(shard, device=device), torch.zeros(shard * W, device=device)
    bucket = torch.zeros(max(1, int(bucket_bytes) // 4), device=device)
    avg = dist.ReduceOp.AVG if dist.get_backend() == 'nccl' else dist.ReduceOp.SUM
    ops = {'all_gather': lambda: dist.all_gather_into_tensor(full, loc), 'reduce_scatter': lambda: dist.reduce_scatter_tensor(loc, full),
           'all_reduce': lambda: dist.all_reduce(bucket, op=avg)}
    out = {}
    for k, f in ops.items():
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        out[k] = e0.elapsed_time(e1) * 1e3 / reps
    out['world'], out['bytes'] = W, {'all_gather': 4 * shard * W, 'reduce_scatter': 4 * shard * W, 'all_reduce': 4 * bucket.numel()}
    MEASURED_FLOOR_US = out
    return out


def choose_shard_or_replicate(world, sweep_ms, sweep_bwd_ms, eik_ms, n_grid, links=1):
    """'shard' or 'replicate' for a W-rank job, from numbers instead of a default: sharding removes (1 - 1/W) of the
    frame-independent kernels of a rank (grid sweep, its sparse backward, the eikonal chain) and adds two collectives (all-gather of sdf,
    reduce-scatter of d(sdf)); the collectives are priced by model_collective_us with the measured floor when there is one.
    Returns (mode, saved_ms, added_ms)."""
    if world <= 1:
        return 'replicate', 0.0, 0.0
    saved = (1.0 - 1.0 / world) * (float(sweep_ms) + float(sweep_bwd_ms) + float(eik_ms))
    added = (model_collective_us('all_gather', 4 * n_grid, world, links) + model_collective_us('reduce_scatter', 4 * n_grid, world, links)) / 1e3
    return ('shard' if saved > added else 'replicate'), saved, added


def model_collective_us(kind, nbytes, world, links=1):
    """microseconds of one collective of `nbytes` (the full buffer) over `world` GPUs of one node.  `links` = 1: a ring bound by ONE link per
    hop (pessimistic on the fully connected 8-GPU node); `links` = world - 1: the direct all-to-all form in which every peer pair uses its own
    link (reduce-scatter + all-gather with 1/W of the buffer per peer)."""
    if world <= 1:
        return 0.0
    bw = XGMI_LINK_GBPS * XGMI_EFFICIENCY * max(1, min(links, world - 1)) * 1e9
    frac = (world - 1) / world
    moved = {'all_reduce': 2 * frac, 'all_gather': frac, 'reduce_scatter': frac}[kind] * nbytes
    lat = (MEASURED_FLOOR_US or {}).get(kind, COLLECTIVE_LATENCY_US)
    return lat + moved / bw * 1e6
