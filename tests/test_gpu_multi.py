"""-m gpu, needs >= 2 GPUs (skipped on the single-GPU test box): the frame-parallel collectives on RCCL with two ranks -- the in-graph
all-gather / reduce-scatter of the sharded SDF sweep (d3h.dist_ops) and the one-bucket gradient all-reduce (Scene.allreduce_grads)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY='0')
    sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    try:
        from d3h import dist_ops
        from d3h.scene import Scene
        dev = f'cuda:{rank}'
        n = 1000                                                   # not a multiple of the shard: the last shard is short
        lo, hi, shard = dist_ops.shard_range(n, rank, world)
        full = torch.arange(n, dtype=torch.float32, device=dev).reshape(n, 1)
        local = full[lo:hi].clone().requires_grad_(True)
        y = dist_ops.gather_shards(local * 2.0, n, shard, rank, world)
        ok = bool(torch.equal(y.detach(), full * 2.0))
        w = (torch.arange(n, dtype=torch.float32, device=dev).reshape(n, 1) % 7 + 1.0) * float(rank + 1)      # rank-dependent upstream gradient
        (y * w).sum().backward()
        want = 2.0 * (torch.arange(n, dtype=torch.float32, device=dev).reshape(n, 1) % 7 + 1.0)[lo:hi] * sum(range(1, world + 1))
        ok = ok and bool(torch.allclose(local.grad, want))
        s = object.__new__(Scene)
        s.shared_params = [torch.nn.Parameter(torch.zeros(5, 3, device=dev)), torch.nn.Parameter(torch.zeros(7, device=dev))]
        s.world = world
        for i, p in enumerate(s.shared_params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        s.allreduce_grads()
        mean = sum(range(1, world + 1)) / world
        ok = ok and bool(torch.allclose(s.shared_params[0].grad, torch.full((5, 3), mean, device=dev)))
        ok = ok and bool(torch.allclose(s.shared_params[1].grad, torch.full((7,), 2 * mean, device=dev))) and s.bucket_bytes == 4 * 22
        dist.barrier()
        # the per-call floor of the step's three collectives at their real sizes (1 MB, 1 MB, 10.1 MB) on this group
        floor = dist_ops.measure_rccl_floor_us(262144, 10126820, dev, reps=20)
        ok = ok and all(0.0 < floor[k] < 5e4 for k in ('all_gather', 'reduce_scatter', 'all_reduce'))
        if rank == 0:
            print('RCCL floor (us) at world', world, {k: round(v, 1) for k, v in floor.items() if k in ('all_gather', 'reduce_scatter', 'all_reduce')}, flush=True)
        q.put((rank, ok, floor if rank == 0 else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the driver runs the multi-GPU bench on an 8-GPU node)')
def test_two_rank_rccl_collectives():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, 29690, q)) for r in range(2)]
    [p.start() for p in ps]
    res = {r: ok for r, ok, _ in (q.get(timeout=300) for _ in ps)}
    [p.join(60) for p in ps]
    assert res == {0: True, 1: True}


def test_one_rank_rccl_runs_the_two_rank_worker_body():
    """the body of the two-rank test on a ONE-rank RCCL group in a spawned process, so that its code path (process-group start-up in a child,
    the in-graph all-gather / reduce-scatter, the bucket all-reduce, the collective-floor measurement) executes on every single-GPU box --
    the two-rank form has never had two GPUs to run on.  The measured floors are printed (bench.py reports them per run)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, 29691, q))
    p.start()
    rank, ok, floor = q.get(timeout=600)
    p.join(120)
    assert (rank, ok) == (0, True) and p.exitcode == 0
    print('one-rank RCCL floors (us):', floor)
    assert floor['world'] == 1 and floor['bytes']['all_reduce'] == 10126820 // 4 * 4
