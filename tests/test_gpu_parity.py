"""-m gpu: the HIP kernels through the C ABI against the oracle / the reference's golden vectors."""
import pytest

import parity_cases as PC

pytestmark = pytest.mark.gpu


def test_gpu_sdf_mlp_forward(gpu):
    PC.check_sdf_mlp_forward(gpu)
    PC.check_sdf_mlp_forward(gpu, x3=True)          # the product's default arithmetic against the same reference outputs


def test_gpu_marching_tets_golden(gpu):
    PC.check_mtets_golden(gpu)


def test_gpu_marching_tets_speculative_equals_exact(gpu):
    PC.check_mtets_speculative(gpu, n=24)


def test_gpu_sdf_mlp_backward(gpu):
    PC.check_sdf_mlp_backward(gpu)
    PC.check_sdf_mlp_backward(gpu, n=1000, sparse_gout=True)
    PC.check_sdf_mlp_backward(gpu, n=777, sparse_gout=True)


def test_gpu_sdf_mlp_eikonal(gpu):
    PC.check_sdf_mlp_eikonal(gpu, n=4096)
    PC.check_sdf_mlp_eikonal(gpu, n=2999, scale=20.0)


def test_gpu_seq_ops(gpu):
    PC.check_seq_ops_golden(gpu)
    PC.check_mesh_api_seq(gpu)
    PC.check_mlp_deform_golden(gpu)
    PC.check_mlp_deform_fused_vs_library(gpu, n=3000)
    PC.check_mlp_deform_fused_vs_library(gpu, n=16385)
    PC.check_mesh_sdf(gpu, n=20000)


def test_gpu_sdf_mlp_deform(gpu):
    PC.check_sdf_mlp_deform(gpu)


def test_gpu_lbs_golden(gpu):
    PC.check_lbs_golden(gpu)


def test_gpu_knn_grid(gpu):
    PC.check_knn_grid(gpu)
    PC.check_knn_grid(gpu, nv=10475, nq=50000, seed=1)


def test_gpu_rasterize(gpu):
    PC.check_rasterize(gpu, res=64)
    PC.check_rasterize(gpu, res=96, big=True, nb=1)


def test_gpu_rasterize_tile_binned(gpu, monkeypatch):
    """the tile-binned rasteriser (large meshes): bit-identical to the wave-per-triangle kernels at 1080 x 1080 (a multiple of no tile) on 60 000
    small triangles, and the oracle checks through it"""
    from d3h import raster
    PC.check_rasterize_binned(gpu, res=1080, n_small=60000)
    PC.check_rasterize_binned(gpu, res=80, n_small=3000)
    monkeypatch.setattr(raster, 'BIN_MIN_TRIS', 1)
    PC.check_rasterize(gpu, res=64)
    PC.check_rasterize_near_plane(gpu)


def test_gpu_interpolate(gpu):
    PC.check_interpolate(gpu, res=48)


def test_gpu_xfm_points(gpu):
    PC.check_xfm_points(gpu)
    PC.check_xfm_points(gpu, n=40000)


def test_gpu_gbuffer(gpu):
    PC.check_gbuffer(gpu, res=40)
    PC.check_gbuffer(gpu, res=256)


def test_gpu_antialias(gpu):
    PC.check_antialias(gpu, res=48)


def test_gpu_texture(gpu):
    PC.check_texture(gpu)


def test_gpu_normals(gpu):
    PC.check_normals_golden(gpu)


def test_gpu_shading_normal(gpu):
    PC.check_shading_normal_golden(gpu)


def test_gpu_image_loss(gpu):
    PC.check_image_loss_golden(gpu)


def test_gpu_ssim(gpu):
    PC.check_ssim_golden(gpu)


def test_gpu_sample_points(gpu):
    PC.check_sample_points(gpu, nv=3000, nf=6000, n=50000)
    PC.check_sample_points(gpu, nv=20000, nf=60000, n=50000)        # 80 000 rows with the padding: five segments of the prefix-sum workgroup


def test_gpu_composite(gpu):
    PC.check_composite(gpu, B=3, H=67, W=129)
    PC.check_first_channels(gpu, B=3, H=67, W=129)
    PC.check_material_grads(gpu)
    PC.check_material_grads(gpu, B=3, H=67, W=129)
    PC.check_seq_losses(gpu)
    PC.check_seq_losses(gpu, B=3, H=65, W=130)


def test_gpu_pixel_losses(gpu):
    PC.check_pixel_losses(gpu, B=2, H=96, W=80)
    PC.check_pixel_losses(gpu, B=1, H=17, W=33, with_ssim=False)


def test_gpu_pixel_losses_ssim_occupancy(gpu):
    PC.check_pixel_losses_ssim_occupancy(gpu)
    PC.check_pixel_losses_ssim_occupancy(gpu, B=2, H=1024, W=1024)


def test_gpu_sdf_reg(gpu):
    PC.check_sdf_reg_golden(gpu)


def test_gpu_texmlp(gpu):
    PC.check_texmlp(gpu, n=5000)


def test_gpu_texmlp_shared_table(gpu):
    PC.check_texmlp_shared_table(gpu, n=3000, passes=3)
    PC.check_texmlp_shared_table(gpu, n=400000, passes=9, vs_oracle=False)


def test_gpu_render_mesh_vs_reference_render(gpu):
    PC.check_render_mesh_golden(gpu)


def test_gpu_gshell_tangents(gpu):
    PC.check_gshell_tangents_golden(gpu)


def test_gpu_rccl_collectives_single_rank(gpu):
    """The in-graph collectives of the frame-parallel step (d3h.dist_ops) and the gradient bucket on the RCCL backend with a
    single-rank group: device tensors, current-stream semantics and autograd plumbing as bench.py uses them at N > 1 (the
    multi-rank arithmetic is pinned by tests/test_distributed_gloo.py)."""
    import os
    import torch
    import torch.distributed as dist
    from d3h import dist_ops
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29688', RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        x = torch.randn(300, 1, device='cuda', requires_grad=True)
        lo, hi, shard = dist_ops.shard_range(300, 0, 1)
        assert (lo, hi, shard) == (0, 300, 384)
        y = dist_ops.gather_shards(x * 2.0, 300, shard, 0, 1)
        assert y.shape == (300, 1) and torch.equal(y, x.detach() * 2.0)
        w = torch.randn(300, 1, device='cuda')
        (y * w).sum().backward()
        assert torch.allclose(x.grad, 2.0 * w)
        from d3h.scene import Scene
        s = object.__new__(Scene)
        s.shared_params = [torch.nn.Parameter(torch.zeros(5, 3, device='cuda')), torch.nn.Parameter(torch.zeros(7, device='cuda'))]
        s.world = 1
        s.shared_params[0].grad = torch.full((5, 3), 2.0, device='cuda')
        s.allreduce_grads()
        # every shared parameter that requires a gradient is a member; one without a local gradient contributes (and receives) zeros
        assert torch.all(s.shared_params[0].grad == 2.0) and not s.shared_params[1].grad.any() and s.bucket_bytes == 88
        dist.barrier()
        # ---- a whole training step through the frame-parallel code path on RCCL: gradient arena opened, sweep through the row-range +
        # all-gather / reduce-scatter ops, the bucket all-reduce(AVG) in place.  The group has ONE rank (the box has one GPU), so every
        # collective is the identity and the step must reproduce the plain single-GPU step on the same seed
        from d3h import gradarena

        torch.manual_seed(0)
        ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4
        sc = Scene(res=128, grid_n=12, n_frames=2, device='cuda', prefit_steps=150, loss_set='full', body_verts=2048, sdf_fn=ell)
        bg = torch.rand(2, 128, 128, 3, device='cuda')

        def tick(parallel):
            sc.world = 2 if parallel else 1          # "has a peer": opens the arena, calls the collectives (AVG over the one real rank)
            sc.FLAGS.sdf_shard = (0, 1) if parallel else None          # the whole grid as rank 0's shard
            torch.manual_seed(1)                     # same surface samples / shading jitter on both sides
            sc._zero_grad()
            r = sc.geometry.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
            r['d3h_total'].backward()
            if parallel:
                sc.allreduce_grads()
                a = sc._arena
                assert sc.geometry.deform.grad.data_ptr() == a.views[a.index[id(sc.geometry.deform)]].data_ptr()
                assert sc.material['kd_ks'].encoder.params.grad.data_ptr() == a.views[a.index[id(sc.material['kd_ks'].encoder.params)]].data_ptr()
            g = [sc.geometry.deform.grad.clone(), sc.material['kd_ks'].encoder.params.grad.clone()] + [p.grad.clone() for p in sc.geometry.sdf_net.parameters()]
            return float(r['d3h_total'].detach()), g
        la, ga = tick(False)
        lb, gb = tick(True)
        assert gradarena.ACTIVE is None
        assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
        for x, y in zip(ga, gb):                      # same parameters, same draws: only the order of the float atomics differs
            assert (x - y).norm() <= 1e-4 * x.norm() + 1e-9, (float((x - y).norm()), float(x.norm()))
        for _ in range(3):                            # and whole optimiser steps through that path stay finite
            r = sc.step()
            assert all(torch.isfinite(v).all() for v in r.values()), r
    finally:
        dist.destroy_process_group()


def test_gpu_render_uv(gpu):
    PC.check_render_uv(gpu)


def test_gpu_fused_adam(gpu):
    PC.check_fused_adam(gpu)


def test_gpu_smplx_pose_kernel(gpu):
    PC.check_smplx_pose_kernel(gpu)


def test_gpu_rasterize_near_plane(gpu):
    PC.check_rasterize_near_plane(gpu)


def test_gpu_composite_antialias_fused(gpu):
    PC.check_composite_antialias_fused(gpu)


@pytest.mark.gpu
def test_gpu_repeated_ticks_of_one_state_agree(gpu):
    """The same tick (same parameters, same random draws) twelve times: every gradient tensor within float-atomic noise of the first run.
    Guards the register-file claim of the bf16-MFMA kernels (csrc/sdf_mlp_x3.h: D3H_X3_CLAIM_SIMD): without it the eikonal chain's
    weight-gradient kernel on the side stream made main-stream kernels sharing its SIMDs return different results in 5-40 % of the ticks
    (a few mesh vertices with a wrong LBS-backward gradient -> deform and every SDF parameter off by per cents)."""
    import torch
    from d3h.scene import Scene
    torch.manual_seed(0)
    ell = lambda x: (((x - torch.tensor([0.0, -0.4, 0.0], device=x.device)) / torch.tensor([0.55, 0.8, 0.45], device=x.device)).norm(dim=-1) - 1.0) * 0.4
    sc = Scene(res=128, grid_n=12, n_frames=2, device='cuda', prefit_steps=150, loss_set='full', body_verts=2048, sdf_fn=ell)
    bg = torch.rand(2, 128, 128, 3, device='cuda')

    def tick():
        torch.manual_seed(1)
        sc._zero_grad()
        r = sc.geometry.tick_init(sc.glctx, sc.target(bg), None, sc.material, sc.loss_fn, 0, None)
        r['d3h_total'].backward()
        g = [sc.geometry.deform.grad.clone()] + [p.grad.clone() for p in sc.geometry.sdf_net.parameters()]
        torch.cuda.synchronize()
        return float(r['d3h_total'].detach()), g
    l0, g0 = tick()
    for it in range(12):
        l, g = tick()
        assert abs(l - l0) <= 1e-5 * abs(l0), (it, l, l0)
        for k, (a, b) in enumerate(zip(g0, g)):
            assert (a - b).norm() <= 1e-4 * a.norm() + 1e-9, (it, k, float((a - b).norm()), float(a.norm()))
