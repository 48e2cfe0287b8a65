"""Frame-parallel data parallelism on CPU (gloo, world_size 2): the one-bucket gradient all-reduce of d3h.scene.Scene.

Checks (i) the bucket all-reduce reproduces the mean of the per-rank gradients for every shared parameter and leaves per-frame pose
rows alone, and (ii) with the emulated kernels: 2 ranks x 1 frame (each sweeping HALF of the tet grid: the sdf all-gather / d(sdf)
all-reduce of d3h.dist_ops) == 1 rank x 2 frames for the shared-parameter gradients
(losses are batch means, so averaging rank gradients equals the 2-frame batch gradient)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker_bucket(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from d3h.scene import Scene
    s = object.__new__(Scene)
    torch.manual_seed(0)
    s.shared_params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2, 2))]
    s.world = world
    s.shared_params.append(torch.zeros(4))                      # a tensor that requires no gradient is not a bucket member
    p0, p1 = s.shared_params[:2]
    p1.grad = torch.full_like(p1, float(rank + 1) * 2)
    if rank == 0:                                               # first step: only rank 0 has a gradient for p0 (an empty garment on rank 1,
        p0.grad = torch.full_like(p0, 3.0)                      # a loss term that switches on later): rank 1 must still take part in its mean
    # third parameter never receives a gradient on any rank (e.g. the cond codes of the init stage): a member contributing zeros
    s.allreduce_grads()
    first = [None if p.grad is None else p.grad.numpy().copy() for p in s.shared_params]
    # second step: a bucket member without a gradient on this step contributes zeros (and gets the mean of the others)
    s.shared_params[0].grad = torch.full((5, 3), 4.0) if rank == 0 else None
    s.shared_params[1].grad = torch.full((7,), 2.0)
    s.allreduce_grads()
    q.put((rank, first, [None if p.grad is None else p.grad.numpy().copy() for p in s.shared_params], s.bucket_bytes))
    dist.destroy_process_group()


def test_bucket_allreduce_is_mean_over_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_bucket, args=(r, 2, 29671, q)) for r in range(2)]
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        rk, first, second, nbytes = q.get(timeout=120)
        res[rk] = (first, second, nbytes)
    [p.join(60) for p in ps]
    for r in range(2):
        first, second, nbytes = res[r]
        assert nbytes == 4 * (15 + 7 + 4)                        # every shared parameter that requires a gradient, whoever produced one
        assert first[3] is None and second[3] is None
        assert not first[2].any() and not second[2].any()
        assert torch.allclose(torch.from_numpy(first[0]), torch.full((5, 3), 1.5)) and torch.allclose(torch.from_numpy(first[1]), torch.full((7,), 3.0))
        assert torch.allclose(torch.from_numpy(second[0]), torch.full((5, 3), 2.0)) and torch.allclose(torch.from_numpy(second[1]), torch.full((7,), 2.0))


def _scene(n_frames, frame_seed, world=1, rank=0):
    from d3h import _lib as L
    from conftest import EMUL_SO
    L._use_emulator_for_tests(EMUL_SO)
    from d3h.scene import Scene
    ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0])) / torch.tensor([0.55, 0.8, 0.45])).norm(dim=-1) - 1.0) * 0.4
    return Scene(res=24, grid_n=5, n_frames=n_frames, device='cpu', prefit_steps=120, loss_set='mask', body_verts=300, sdf_fn=ell,
                 flags_hook=lambda F: (setattr(F, 'prefit_with_library_path', True), setattr(F, 'use_eikonal', False)), frame_seed=frame_seed, dist_world=world, dist_rank=rank)


def _worker_equiv(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    for p in (ROOT, os.path.join(ROOT, 'd3human-code_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(4)          # two workers share the container's 8 cores
    torch.manual_seed(0)
    sc = _scene(1, 1234 + rank, world, rank)
    # both ranks must use the targets/background of "their" frame of the 2-frame reference run
    ref = torch.load(os.environ['D3H_REF_PT'])
    sc.all_img, sc.all_normal = ref['all_img'][rank:rank + 1], ref['all_normal'][rank:rank + 1]
    tgt = sc.target(ref['bg'][rank:rank + 1])
    if os.environ.get('D3H_TEST_SHARD', '1') == '1':          # grid n=5: 216 vertices -> rank 0 sweeps 128 of them, rank 1 the other 88
        sc.enable_sweep_sharding()       # both ranks hold identical parameters here (same seed, deterministic CPU pre-fit)
        assert sc.FLAGS.sdf_shard == (rank, world)
    sc._zero_grad()
    r = sc.geometry.tick_init(sc.glctx, tgt, None, sc.material, sc.loss_fn, 0, None)
    r['msk_loss'].backward()
    sc.allreduce_grads()
    q.put((rank, sc.geometry.deform.grad.numpy().copy(), sc.geometry.sdf_net.net[0].weight.grad.numpy().copy()))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_one_frame_equals_one_rank_two_frames(emul_lib, tmp_path):
    torch.manual_seed(0)
    sc = _scene(2, 1234)
    bg = torch.rand(2, 24, 24, 3)
    sc._zero_grad()
    r = sc.geometry.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
    r['msk_loss'].backward()
    ref_deform, ref_w = sc.geometry.deform.grad.clone(), sc.geometry.sdf_net.net[0].weight.grad.clone()
    path = str(tmp_path / 'ref.pt')
    torch.save({'all_img': sc.all_img, 'all_normal': sc.all_normal, 'bg': bg}, path)
    os.environ['D3H_REF_PT'] = path
    from d3h import _lib as L
    L._lib, L._emulated = None, False
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_equiv, args=(rk, 2, 29673, q)) for rk in range(2)]
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        rk, gd, gw = q.get(timeout=800)
        res[rk] = (torch.from_numpy(gd), torch.from_numpy(gw))
    [p.join(60) for p in ps]
    assert ref_deform.abs().max() > 0
    for rk in range(2):
        gd, gw = res[rk]
        assert (gd - ref_deform).abs().max() < 1e-4 * ref_deform.abs().max() + 1e-7
        assert (gw - ref_w).abs().max() < 1e-4 * ref_w.abs().max() + 1e-7
