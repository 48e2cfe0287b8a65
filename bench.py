#!/usr/bin/env python3
"""bench.py -- train iterations/sec of the D3-Human init-stage render-and-fit step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one iteration of the reference's loop body (train.py:679-790) on one batch of synthetic frames:
SDF MLP sweep over all tet-grid vertices -> G-Shell marching tets -> per-frame SMPL-X LBS -> rasterize / interpolate / texture-MLP /
antialias -> mask + normal + SSIM (+ sdf_reg + eikonal) losses -> backward -> Adam steps -> clamp -> stream sync.
N = 1 workload = BASELINE.json configs[2] (the configuration the metric is quoted on): 4-frame batch, tet-res 128, 1024^2.
N > 1: frame-parallel weak scaling -- every rank runs the same per-GPU batch on its own frames and ONE flat fp32 bucket of the
shared-parameter gradients is all-reduced over RCCL per step; the SDF sweep over the tet grid (identical on every rank) is sharded N ways
with an all-gather of the values / all-reduce of their gradients; value = N*K / T (iterations of a 4-frame batch per second, whole job).

Prints ONE JSON line (rank 0) with the `roofline` of the dominant kernel (the fused SDF query, fp32 MFMA bound) measured live with
HIP events on the launch stream, and a `cpu_baseline` (the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))

FLOP_PER_POINT_FWD = 826880          # 2*(39*256 + 3*256^2 + 295*256 + 2*256^2 + 256)
BYTES_PER_POINT_FWD = 16             # 12 B in + 4 B out (algorithmic)
MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact f32
# HBM bytes per sdf_mlp_fwd_kernel launch at 262 144 points WITH the activation save of the training step, from the PMC counters
# (profiles/r1_pmc_fetch_write_bench_config3_v2.csv: FETCH_SIZE 75 834 KiB -- doubled per the gfx950 correction for wide coalesced
# reads -- + WRITE_SIZE 1 841 152 KiB).  1.88 GB of it is the deliberate tile-packed activation store for the backward pass.
PMC_TRAFFIC_BYTES = {262144: (2 * 75834 + 1841152) * 1024}


def cpu_baseline(grid_n, budget_pts=65536):
    """oracle (torch CPU restatement of the reference) timed on the host cores -- a reported baseline, not the target"""
    import torch
    from oracle import sdf_mlp as OMLP, marching_tets as OMT
    from d3h import synth
    torch.manual_seed(0)
    dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
    sd = {}
    for li, (i, o) in enumerate(dims):
        l = torch.nn.Linear(i, o)
        sd[f'net.{2 * li}.weight'], sd[f'net.{2 * li}.bias'] = l.weight, l.bias
    verts, tets = synth.kuhn_grid(grid_n)
    v = torch.from_numpy(verts)
    n_all = v.shape[0]
    n = min(budget_pts, n_all)
    x = v[:n].clone().requires_grad_(True)
    t0 = time.time()
    y = OMLP.mlp_forward(x, sd)
    y.sum().backward()
    t_sdf = (time.time() - t0) * (n_all / n)                      # the sweep is linear in the point count
    sdf = synth.body_sdf(v)
    msdf = (torch.rand(n_all) - 0.01).clamp(-1, 1)
    t0 = time.time()
    OMT.gshell_tets(v, sdf, msdf, torch.from_numpy(tets))
    t_mt = time.time() - t0
    return {'value': 1.0 / (t_sdf + t_mt), 'unit': 'iters/s (upper bound)', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle SDF MLP fwd+bwd on {n} of {n_all} grid points (scaled linearly to the full sweep: {t_sdf:.2f} s) + oracle '
                      f'marching tets on the full n={grid_n} grid ({t_mt:.2f} s); LBS/render/loss/optimizer stages NOT included, so this '
                      f'is an upper bound of the CPU iteration rate'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=3, help='BASELINE.json config (1-based): 2 = res64/512^2/1 frame/mask, 3 = res128/1024^2/4 frames/full (the metric), 5 = split stage; 6 = seq stage (extra)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--prefit', type=int, default=300)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the product has no CPU path)')
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(backend='nccl', init_method='env://')         # "nccl" is RCCL on ROCm
    dev = f'cuda:{local}'

    from d3h import scene, sdf_mlp
    if args.config == 2:
        cfg = dict(res=512, grid_n=32, n_frames=1, loss_set='mask')
        name = 'config2: 1 frame, tet-res 64 (Kuhn n=32: 35937 verts / 196608 tets), 512x512, mask loss only'
    elif args.config == 5:
        cfg = dict(res=1024, grid_n=63, n_frames=4, loss_set='split')
        name = ('config5 (per GPU): dual garment+body pass (hmSDF_Tets cloth + body, tick_split x2 per iteration), tet-res 128, 1024x1024, '
                '4 frames; loss stack of tick_split with the MSE+cos normal term (no LPIPS/MobileNet weights offline)')
    elif args.config == 6:
        cfg = dict(res=1024, grid_n=63, n_frames=1, loss_set='seq')
        name = ('seq stage (not a BASELINE config; SURVEY 8(f) rank 1): fixed-topology body + garment mesh, MLP_deform offsets, LBS, '
                'render_mask, tick_seq loss stack (masks, image, material regularisers, normal MSE+cos, Laplacian, normal consistency, '
                'collision), 1024x1024, 1 frame')
    else:
        cfg = dict(res=1024, grid_n=63, n_frames=4, loss_set='full')
        name = 'config3: 4-frame batch, tet-res 128 (Kuhn n=63: 262144 verts / 1500282 tets), 1024x1024, mask+normal+SSIM+sdf_reg+eikonal'
    sc = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, dist_world=world, dist_rank=rank,
                     frame_seed=1234 + rank * cfg['n_frames'], **cfg)
    if world > 1:      # identical shared parameters on every rank
        for p in sc.shared_params:
            dist.broadcast(p.data, src=0)
        sc.enable_sweep_sharding()       # each rank sweeps 1/N of the tet grid; sdf all-gathered, d(sdf) all-reduced (d3h/dist_ops.py)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = {'split': sc.step_split, 'seq': sc.step_seq}.get(cfg['loss_set'], sc.step)
    for _ in range(args.warmup):
        step()
    sync()
    sdf_mlp.TIMING = []                      # HIP events around every fused SDF-query forward launch (on the launch stream)
    t0 = time.time()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.time() - t0
    ev = sdf_mlp.TIMING
    sdf_mlp.TIMING = None
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    # roofline of the dominant kernel (sdf_mlp_fwd_kernel over the full grid)
    # (N > 1: every rank sweeps its 1/N slice of the grid, so the largest launches are n_grid / N points)
    n_grid = max((n for _, _, n in ev), default=sc.geometry.verts.shape[0])
    durs = [a.elapsed_time(b) for a, b, n in ev if n == n_grid]
    avg_ms = sum(durs) / max(len(durs), 1)
    tflops = FLOP_PER_POINT_FWD * n_grid / (avg_ms * 1e-3) / 1e12 if durs else None
    roof = {'kernel': 'sdf_mlp_fwd_kernel<false, %d>' % (0 if (n_grid + 127) // 128 >= 1024 else 1), 'bound': 'mfma', 'achieved': tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': (tflops / MFMA_F32_PEAK_TFLOPS) if tflops else None, 'traffic': PMC_TRAFFIC_BYTES.get(n_grid),
            'traffic_note': 'bytes/launch from rocprofv3 PMC (profiles/r1_pmc_fetch_write_bench_config3_v2.csv), incl. 1.88 GB saved activations', 'launch_ms': avg_ms, 'launches': len(durs), 'points_per_launch': int(n_grid),
            'algorithmic_GBps': (BYTES_PER_POINT_FWD * n_grid / (avg_ms * 1e-3) / 1e9) if durs else None}
    md = sc.geometry.last_mesh_dict
    if 'imesh' not in md:
        md = {'imesh': md['all_mesh']}
    out = {'metric': 'train iters/sec @ tet-res 128, 1024^2 render; 1/2/4/8 MI355X', 'value': world * args.steps / dt, 'unit': 'iters/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': name, 'frames_per_gpu': cfg['n_frames'], 'mesh_verts': int(md['imesh'].v_pos.shape[0]),
                      'mesh_faces': int(md['imesh'].t_pos_idx.shape[0]), 'watertight_render': True,
                      'buffers': 'loss-consumed only', 'parallelism': f'frame-parallel dp{world}',
                      'loss': {k: float(v) for k, v in sc.last.items()}},
           'roofline': roof}
    if not args.no_cpu_baseline and world == 1:          # the CPU baseline is measured once, on the single-GPU run
        out['cpu_baseline'] = cpu_baseline(cfg['grid_n'])
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
