#!/usr/bin/env python3
"""bench.py -- train iterations/sec of the D3-Human init-stage render-and-fit step on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...`: WORLD_SIZE is set and must equal N) or bench.py does it itself: with WORLD_SIZE unset the parent -- before any GPU call --
starts that very command as a CHILD process (never exec), relays rank 0's JSON line and exits with the child's code; a rank that dies
gives a non-zero exit and no line.

One "step" = one iteration of the reference's loop body (train.py:679-790) on one batch of synthetic frames:
SDF MLP sweep over all tet-grid vertices -> G-Shell marching tets -> per-frame SMPL-X LBS -> rasterize / interpolate / texture-MLP /
antialias -> mask + normal + SSIM (+ sdf_reg + eikonal) losses -> backward -> Adam steps -> clamp -> stream sync.
N = 1 workload = BASELINE.json configs[2] (the configuration the metric is quoted on): 4-frame batch, tet-res 128, 1024^2.

N > 1, default ("weak"): frame-parallel -- every rank runs the same per-GPU batch on its own frames and ONE flat fp32 bucket of the
shared-parameter gradients (the step's gradient arena, d3h/gradarena.py) is all-reduced over RCCL per step; the frame-INDEPENDENT work
of the step is split over the ranks, not replicated (d3h/dist_ops.py): 1/N of the SDF sweep over the tet grid per rank (sdf all-gathered,
d(sdf) reduce-scattered: two 1 MB collectives) and 50 000 / N eikonal samples per rank.  value = N*K / T (rank-iterations per second, the
bench contract); `optimizer_steps_per_s` (= K / T) and `frames_per_s` stand beside it.
  --replicate       N > 1: every rank repeats the whole sweep and all 50 000 samples (the round-1..3 default): one collective per step.
  --frames-total F  BASELINE configs[3] ("strong"): F frames in total, F / N per GPU (8 frames on 8 GPUs = one per GPU);
                    N = 1 runs the same F-frame batch on one GPU; value = K / T.
  --as-rank-of W    ONE GPU, no process group: this process runs exactly the work of rank --rank-index (default W // 2) of a W-rank job
                    (d3h.dist_ops virtual-rank mode: the sdf shards of the other ranks come from a resident full sweep, learning rates at
                    zero so they stay exact; every kernel and every byte of glue of a real rank's step runs) and prints its step time.
Without --frames-total the same invocation ALSO times configs[3] (8 frames in total over the N ranks) after the headline run and
reports it as `config.config4_frames_total_8`; the N = 1 run additionally measures one virtual rank of W = 2, 4, 8 for both forms and
emits `config.predicted_scaling` (rank time + the xGMI model of the collectives).

Prints ONE JSON line < 4 KB as the last line of stdout (rank 0; short_line()): the contract's keys, `config` (workload, frames, faces,
the 12-buffer and exact-f32 rates, for N > 1 world size / backend / collective microseconds), the `roofline` of the dominant kernel (the
fused SDF query on the matrix pipe; HIP events on its launch stream inside the timed region, csrc/timing.hip) and `cpu_baseline` (the
oracle tick of configs[1] on the host cores + the parity of that very run as scalars).  The FULL record -- `rooflines` of the other
heavy kernels, `predicted_scaling`, the whole parity report, per-stage CPU seconds -- is written to bench_detail.json beside this script
($D3H_BENCH_DETAIL overrides; emit()), never to stdout: the round-4 line had grown to 23 KB and the driver could not parse it.
"""
import argparse
import ctypes
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
MFMA_BF16_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_16x16x32_bf16: 16 cycles per 16x16x32 block)
# The forward / tangent / data-backward sweeps of the SDF network run their fp32 GEMMs on the bf16 pipe with every operand split into three
# bf16 numbers and SIX bf16 products per fp32 product (csrc/sdf_mlp_x3.h; fp32 accumulate, fp32-level error): their roof is the bf16 peak
# divided by six, in algorithmic (fp32) FLOP/s.  D3H_SDF_X3=0 runs the exact-f32 MFMA kernels, priced against the f32 matrix peak.
X3_PRODUCTS = 6
# Round 6: the forward-type sweeps (the grid sweep, the eikonal forward, the recompute of the sparse backward: kernel ids 0 and 27) split their
# operands into TWO fp16 planes (value + residual scaled by 2^11) and take THREE fp16 products per fp32 product (csrc/sdf_mlp_x3.h "h2";
# D3H_SDF_H2=0 puts them back on the bf16 x 3 split).  fp16 and bf16 MFMAs run at the same rate: their roof is the dense peak divided by three.
H2_PRODUCTS = 3
H2_KERNEL_IDS = (0, 27)          # + the tangent sweep (2), the data-backward sweeps (1, 3, 5) and the weight-gradient GEMMs (4, 6) when their
#                                  switches are on (d3h.sdf_mlp.H2_JVP / H2_BWD, D3H_DW_H2): see h2_kernel_ids()


def h2_kernel_ids(sm):
    ids = set(H2_KERNEL_IDS)
    if getattr(sm, 'H2_JVP', False):
        ids.add(2)
    if getattr(sm, 'H2_BWD', False):
        ids.update((1, 3, 5))
        if os.environ.get('D3H_DW_H2', '1') != '0':
            ids.update((4, 6))
    return ids
X3_KERNEL_IDS = (0, 1, 2, 3, 5, 27) + ((4, 6) if os.environ.get('D3H_DW_X3', '1') == '1' else ())      # (4, 6: the hidden-layer weight-gradient GEMMs)
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# HBM bytes per grid-sweep launch at 262 144 points WITH the activation save of the training step, from the PMC counters (separate
# FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 correction for wide coalesced reads).  1.88 GB of it is the deliberate
# tile-packed activation store for the backward pass.  bf16 x 3 kernel (profiles/r4_pmc_fetch_write.csv): FETCH 216 263 KiB -- the 2.56 MB
# weight pack is re-streamed L2 -> LDS by every 128-point tile and a few per cent of those reads miss next to the 1.88 GB store stream --
# + WRITE 1 874 440 KiB.  Exact-f32 kernel (profiles/r3_pmc_fetch_write.csv): FETCH 8 520 KiB + WRITE 1 836 034 KiB.
PMC_TRAFFIC_BYTES = {262144: (2 * 8520 + 1836034) * 1024}
PMC_TRAFFIC_BYTES_X3 = {262144: int((2 * 216262.8 + 1874440.2) * 1024)}
# the same kernel WITHOUT the activation save (round 5: the training sweep; profiles/r5_pmc_fetch_write.csv)
# the fp16 x 2 sweep (round 6; profiles/r6_pmc_fetch_write.csv): FETCH 8 628 KiB (x2) + WRITE 10 240 KiB (sdf + deformed points, unchanged from the
# bf16 x 3 kernel): 28.2 MB = 6.7 x the algorithmic 4.2 MB -- the 1.7 MB fp16 pack re-streamed by 2 048 tiles mostly hits L2 (round 5: 72.8 MB)
PMC_TRAFFIC_BYTES_H2_NOSAVE = {262144: int((2 * 8628.4 + 10240.0) * 1024)}
PMC_TRAFFIC_BYTES_X3_NOSAVE = {262144: int((2 * 30434.8 + 10240.0) * 1024)}          # FETCH 30 435 KiB (x2), WRITE 10 240 KiB: 72.8 MB = 17 x the algorithmic 4.2 MB (the weight pack re-streamed by 2 048 tiles, a few per cent of it missing L2); round 4 with the store: 2 362 MB

# kernel ids of csrc/d3h_common.h (D3H_KT_*) -> (name, bound, algorithmic work per unit, unit, note).  FLOP figures count the GEMMs of
# the network shape; byte figures are the compulsory HBM traffic of the pass.
KT = {
    0: ('sdf_mlp_fwd_kernel', 'mfma', FLOP_PER_POINT_FWD, 'point', 'PE + 8 layers forward'),
    1: ('sdf_mlp_bwd_data_kernel<false> (dense)', 'mfma', FLOP_PER_POINT_FWD, 'point', 'dH_{l-1} = W_l^T dZ_l through all 8 layers incl. d(encoding): same GEMM sizes as the forward'),
    2: ('sdf_mlp_fwd_kernel<true> (tangent sweep)', 'mfma', FLOP_PER_POINT_FWD, 'point', 't_l = s_l * (W_l t_{l-1}): the forward GEMMs once more'),
    3: ('sdf_mlp_bwd_data_kernel<true> (injected reverse)', 'mfma', FLOP_PER_POINT_FWD - 2 * 39 * 256, 'point', 'reverse sweep without d(encoding)'),
    4: ('sdf_mlp_bwd_dw_layers_kernel (dual source)', 'mfma', 6 * 2 * 2 * 256 * 256, 'point', 'six 256x256 weight-gradient GEMMs, two sources each (dZ x t and dZ^ x h)'),
    5: ('sdf_mlp_bwd_data_kernel<false> (sparse sweep)', 'mfma', FLOP_PER_POINT_FWD, 'active point', 'only 16-point tiles with a non-zero upstream gradient'),
    6: ('sdf_mlp_bwd_dw_layers_kernel (sparse sweep)', 'mfma', 6 * 2 * 256 * 256, 'active point', 'six 256x256 weight-gradient GEMMs over the active tiles'),
    7: ('texmlp_bwd_mlp_kernel', 'mfma', 2 * (2 * 1536 + 3 * 1536), 'covered pixel', 'forward recompute + backward mat-vecs (VALU) + three weight-gradient outer products (MFMA) of the 10-32-32-6 MLP'),
    8: ('texmlp_bwd_kernel<1> (grid encoding)', 'hbm', 40 + 12 + 12 + 16, 'covered pixel', 'd(encoding) in, position gradient out; the 4.3 MB tables and their gradient stay in L2'),
    9: ('gbuffer_bwd_kernel', 'hbm', None, 'pixel', '16 B raster per pixel + 72 B of attribute gradients per covered pixel'),
    10: ('texmlp_fwd_mfma_kernel', 'hbm', 12 + 24, 'covered pixel', 'position in, 6 channels out; the 40 table gathers per pixel hit the L2-resident 4.3 MB tables (gather-latency bound, not MFMA: 3 kFLOP per pixel)'),
    11: ('aa_fwd_kernel', 'hbm', 8, 'float', 'image in + image out (+ 16 B raster per pixel)'),
    12: ('raster_tris_kernel + raster_resolve_kernel (+ z-buffer fill)', 'hbm', 8 + 8 + 16 + 16, 'pixel', 'z-buffer fill + read, rast and rast_db out (the 64-bit atomicMin traffic of the covering triangles comes on top)'),
    13: ('aa_hash_build_kernel (+ table fills)', 'hbm', None, 'triangle', 'edge -> opposite-vertex hash of the current topology; latency / atomics'),
    14: ('composite_fwd_kernel', 'hbm', 8, 'float', 'layer buffers in, composited channel-concatenated image out (+ 16 B raster per pixel)'),
    15: ('composite_bwd_kernel', 'hbm', 8, 'float', 'd(image) in, d(layer buffers) out'),
    16: ('pixel_losses_fwd_kernel', 'hbm', 4, 'float', 'stacked image in (+ 28 B of references and 24 B of SSIM planes per pixel)'),
    17: ('pixel_losses_bwd_kernel', 'hbm', 8, 'float', 'stacked image in, d(stacked) out'),
    18: ('ssim_fwd_slide_kernel', 'hbm', 8 + 20, 'plane pixel', 'two planes in, five moment-gradient planes out'),
    19: ('ssim_bwd_slide_kernel', 'hbm', 20 + 8 + 4, 'plane pixel', 'five moment-gradient planes + two planes in, one gradient plane out'),
    20: ('mt_count_edges + mt_count_tets + 2 x mt_scan', 'hbm', 16 + 8 + 1, 'tet', 'int32 tets + (amortised) edge list in, case codes out: SURVEY 8(d) 24-48 MB per call at 1.5 M tets'),
    21: ('mt_emit_verts + mt_emit_faces_wt + mt_scan', 'hbm', 24 + 1, 'tet', 'per-tet edge ids + case codes in; vertices / faces out (small)'),
    22: ('lbs_fwd_kernel', 'hbm', 12 + 12 + 220, 'vertex x frame', 'point in / out + the 55-weight row of its nearest template vertex (L2-resident rows: latency-bound)'),
    23: ('lbs_bwd_kernel', 'hbm', 12 + 12 + 12 + 220, 'vertex x frame', 'as the forward + the upstream gradient'),
    24: ('aa_bwd_kernel', 'hbm', 8, 'float', 'd(image out) in, d(image in) out (+ 16 B raster per pixel)'),
    25: ('gbuffer_fwd_kernel', 'hbm', 16 + 40, 'pixel', '16 B raster in, ~10 attribute channels out'),
    26: ('raster_bwd_kernel', 'hbm', 16 + 16, 'pixel', 'rast + d(rast) in; vertex-position atomics of the covered pixels on top'),
    27: ('sdf_mlp_fwd_x3_kernel<false, 1, true> (recompute of the gathered active points)', 'mfma', FLOP_PER_POINT_FWD, 'active point', 'the training sweep runs without the activation save; the sparse backward recomputes the activations of the ~3 % of the grid it visits'),
}


SDF_KT_MASK = 0x7F | (1 << 27)  # kernel ids 0-6 and 27 of csrc/d3h_common.h: the SDF-network launches

# Kernels whose limiter is NOT a bandwidth or FLOP roof: their `rooflines` entry carries the counters that say what binds them instead of a
# fraction of 8 TB/s (profiles/r4_pmc_image_space_kernels.txt: rocprofv3 --pmc, mean per launch of the serialised config-3 step, 4 x 1024^2,
# ~1.06 M covered pixels; wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES)
LIMITER = {
    10: ('l2-gather', '40 corner gathers of 8 B per covered pixel from the 4.3 MB tables: 44.0 M TCP reads per launch (41 per covered pixel), 4.22 M L2 '
                      'requests at a 0.77 hit rate (TCC_HIT 3.23 M), wait share 0.57 -- gather latency on L2-resident data, not HBM bytes'),
    8: ('fabric-atomics', '0.785 M memory-side float atomics per launch (TCC_EA0_ATOMIC = TCC_ATOMIC: every one goes out to the fabric) after the '
                          'segmented wave scans (3.66 M lane atomics at the TCP), wait share 0.72'),
    9: ('fabric-atomics', '0.936 M memory-side float atomics per launch (vertex-attribute gradients, one per run of lanes on a triangle and channel), '
                          '7.8 M lane atomics at the TCP, wait share 0.91'),
    23: ('fabric-atomics', 'pose / point gradients through per-frame segmented sums; 13 k lane atomics, wait share 0.79: latency of one dependent gather '
                           'chain per vertex (55-weight row of its nearest template vertex)'),
    26: ('fabric-atomics', 'vertex-position atomics of the covered pixels (segmented per triangle run) on top of 32 B / pixel'),
}


SHORT_LINE_LIMIT = 4096          # the driver keeps a bounded tail of stdout: the result line must fit in it with room to spare


def short_line(out):
    """The ONE stdout line of a run: the bench contract's keys, `config` reduced to what names the workload, the one `roofline` dict without
    prose and `cpu_baseline` as scalars.  Everything else (`rooflines`, `predicted_scaling`, the full parity report, per-stage CPU seconds)
    goes to the detail file (emit)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'optimizer_steps_per_s', 'frames_per_s')
    s = {k: out[k] for k in keep if k in out}
    c = out.get('config') or {}
    sc = {'workload': str(c.get('workload', ''))[:200]}
    for k in ('frames_per_gpu', 'mesh_faces', 'parallelism', 'all_12_buffers_iters_per_s', 'exact_f32_mfma_iters_per_s', 'graph_replay',
              'world_size', 'backend', 'rccl_ranks_seen', 'as_rank_of', 'rank_index', 'mode', 'frames_per_rank'):
        if c.get(k) is not None:
            sc[k] = c[k] if not isinstance(c[k], str) else c[k][:120]
    if isinstance(c.get('collective'), dict):
        sc['collective'] = {k: c['collective'].get(k) for k in ('bytes', 'avg_us', 'calls', 'extra_collectives_per_step')}
    if isinstance(c.get('config4_frames_total_8'), dict):
        sc['config4_frames_total_8'] = {k: c['config4_frames_total_8'].get(k) for k in ('value', 'ms_per_step', 'frames_per_gpu', 'collective_avg_us')
                                        if c['config4_frames_total_8'].get(k) is not None}
    if isinstance(c.get('other_mode'), dict):
        sc['other_mode'] = {k: c['other_mode'].get(k) for k in ('mode', 'value', 'ms_per_step', 'collectives_per_step')}
    if isinstance(c.get('rccl_floor_us'), dict):
        sc['rccl_floor_us'] = {k: c['rccl_floor_us'].get(k) for k in ('all_gather', 'reduce_scatter', 'all_reduce', 'world')}
    s['config'] = sc
    r = out.get('roofline')
    if isinstance(r, dict):
        s['roofline'] = {k: r.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launch_ms', 'points_per_launch',
                                               'products_per_fp32_product', 'frac_of_bf16x3_roof') if k in r}
    else:
        s['roofline'] = None
    cb = out.get('cpu_baseline')
    if isinstance(cb, dict):
        s['cpu_baseline'] = {k: (cb[k] if not isinstance(cb[k], str) else cb[k][:300]) for k in
                             ('value', 'unit', 'cores', 'kind', 'sample', 'config', 'gpu_same_config_iters_per_s', 'parity_summary') if k in cb}
    if out.get('detail_file'):
        s['detail_file'] = out['detail_file']
    line = json.dumps(s)
    if len(line) >= SHORT_LINE_LIMIT:          # never let prose push the numbers out of the driver's window
        s['cpu_baseline'] = {k: v for k, v in (s.get('cpu_baseline') or {}).items() if k != 'sample'} or None
        s['config']['workload'] = s['config']['workload'][:80]
        line = json.dumps(s)
    assert len(line) < SHORT_LINE_LIMIT, len(line)
    return line


_RESULT_FD = None


def claim_stdout():
    """From now on file descriptor 1 belongs to the result line alone.  Libraries print there too -- RCCL's version banner leaves its stdio buffer
    at process exit, i.e. AFTER the result line (profiles/r5_bench_config3.json of the first round-5 runs: five banner lines behind the JSON) --
    and a driver that parses the LAST line of stdout would read the banner.  fd 1 is duplicated for emit(), then pointed at stderr for everybody
    else (C stdio of every library included), python's sys.stdout likewise."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = sys.stderr


def write_result_line(line):
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        sys.stderr.flush()
        os.write(_RESULT_FD, (line + '\n').encode())


def emit(out):
    """Full record -> the detail file ($D3H_BENCH_DETAIL, default bench_detail.json beside this script; gpurun_out/ gets a copy when it
    exists), then the short result line as the LAST line of stdout."""
    path = os.environ.get('D3H_BENCH_DETAIL', os.path.join(ROOT, 'bench_detail.json'))
    try:
        with open(path, 'w') as fh:
            json.dump(out, fh, indent=1)
        out['detail_file'] = os.path.relpath(path, ROOT)
        gdir = os.path.join(ROOT, 'gpurun_out')
        if os.path.isdir(gdir) and 'D3H_BENCH_DETAIL' not in os.environ:
            with open(os.path.join(gdir, 'bench_detail.json'), 'w') as fh:
                json.dump(out, fh, indent=1)
    except OSError as e:
        sys.stderr.write(f'bench.py: could not write the detail file {path}: {e}\n')
    write_result_line(short_line(out))


def collect_kernel_timing(lib):
    n = int(lib.d3h_timing_read(None, None, None, ctypes.c_int64(0)))
    if n <= 0:
        return []
    ids, units, ms = (ctypes.c_int * n)(), (ctypes.c_int64 * n)(), (ctypes.c_float * n)()
    lib.d3h_timing_read(ids, units, ms, ctypes.c_int64(n))
    return [(int(ids[i]), int(units[i]), float(ms[i])) for i in range(n)]


def cpu_baseline_config3_scaled(grid_n_full, res_full, frames_full, samples_full=50000):
    """The oracle chain (oracle/tick.py: the pinned CPU restatement of the reference's tick_init) timed stage by stage on the host cores on
    a bounded sample, each stage scaled to the full workload by its own size law.  A reported baseline, not the target."""
    import numpy as np
    import torch
    from oracle import tick as OTK, render as ORD, sdf_mlp as OMLP, marching_tets as OMT, lbs as OL, image_ops as OI, raster as OR
    from d3h import synth
    torch.manual_seed(0)
    n_s, res_s, eik_s, body_v = 48, 512, 50000, 4096
    verts, tets = (torch.from_numpy(a) for a in synth.kuhn_grid(n_s))
    m = synth.make_body_model(n_verts=body_v, seed=0, n_shape=10, n_expr=5)
    body = {k: torch.from_numpy(v) for k, v in m.items() if k != 'posedirs'}
    net = torch.nn.ModuleList([torch.nn.Linear(i, o) for i, o in [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]])
    sd = {}
    for li, l in enumerate(net):
        sd[f'net.{2 * li}.weight'], sd[f'net.{2 * li}.bias'] = l.weight, l.bias
    # no pre-fit: the network keeps its default initialisation (its cost does not depend on the weights) and the VALUES the extraction sees
    # are the analytic humanoid SDF, spliced in with a detached correction so that the gradient still flows through the network
    analytic = synth.body_sdf(verts).reshape(-1, 1)
    leaf = lambda t: t.clone().requires_grad_(True)
    z3, z45 = torch.zeros(1, 3), torch.zeros(1, 45)
    bp0 = torch.zeros(1, 63); bp0[:, 2] = torch.pi / 36; bp0[:, 5] = -torch.pi / 36
    A0 = OL.pose_transforms(body, OL.full_pose(z3, bp0, z3, z3, z3, z45, z45), OL.joints_from_shape(body, torch.zeros(1, 10), torch.zeros(1, 5)))[0]
    tmpl = OL.blend_apply(body['v_template'], A0, body['weights'], False)
    mv, mvp, campos = synth.camera(res_s)
    from oracle import texmlp as OT
    table = (torch.rand(2 * OT.grid_layout()[1]) * 2 - 1) * 1e-4
    st = {'verts': verts, 'indices': tets, 'deform': leaf(torch.zeros_like(verts)), 'msdf': leaf(torch.ones(verts.shape[0])), 'max_disp': 1.0 / (2 * n_s) / 2.1,
          'sd': sd, 'body': body, 'tmpl': tmpl, 'A0': A0, 'shape': torch.zeros(1, 10), 'expr': torch.zeros(1, 5), 'root_pose': torch.zeros(1, 3),
          'body_pose': synth.poses(1), 'jaw_pose': torch.zeros(1, 3), 'trans': leaf(torch.zeros(1, 3)), 'mvp': torch.from_numpy(mvp)[None],
          'campos': torch.from_numpy(campos)[None], 'res': (res_s, res_s),
          'material': {'table': leaf(table), 'w1': leaf(torch.randn(32, 10) * 0.3), 'w2': leaf(torch.randn(32, 32) * 0.2), 'w3': leaf(torch.randn(6, 32) * 0.2),
                       'bbox': (0.6, 0.6, 0.2, -0.8, -1.2, -0.2), 'omin': [0, 0, 0, 0, 0.001, 0], 'omax': [1, 1, 1, 0, 1, 1]},
          'all_img': torch.rand(1, res_s, res_s, 4).round(), 'all_normal': torch.nn.functional.normalize(torch.randn(1, res_s, res_s, 3), dim=-1),
          'background': torch.rand(1, res_s, res_s, 3), 'iteration': 10, 'n_iter': 2001, 'sdf_regularizer': 0.2, 'eikonal_scale': None,
          'ssim_weight': 1.0, 'loss_set': 'full'}
    edges = OTK.all_edges(tets)                 # static per grid (hmsdf.py:382-388 builds it once)
    T = {}
    t0 = time.time()
    v_def = st['verts'] + st['max_disp'] * st['deform']
    sdf = OMLP.mlp_forward(v_def, sd)
    T['sdf_sweep_fwd'] = time.time() - t0
    sdf = sdf + (analytic - sdf).detach()
    t0 = time.time()
    mt = OMT.gshell_tets(v_def, sdf, st['msdf'], tets)
    T['marching_tets'] = time.time() - t0
    t0 = time.time()
    A = OTK.frame_transforms(st, [0])
    posed, _, _ = OL.lbs_forward(mt['verts'], tmpl, body['weights'], A0, A[0], st['trans'][0])
    T['lbs'] = time.time() - t0
    st['sampled_pts'] = OTK.surface_samples(posed.detach(), mt['faces'], eik_s)
    t0 = time.time()
    b = ORD.render_mesh(posed[None], mt['verts'], mt['faces'], posed[None], st['mvp'], st['campos'], st['res'], st['material'], background=st['background'],
                        msdf=mt['msdf'], buffers=('shaded', 'geometric_normal', 'msdf_image'))
    T['render'] = time.time() - t0
    t0 = time.time()
    e = OTK.eikonal(st, st['sampled_pts'], 10)
    T['eikonal_fwd'] = time.time() - t0
    t0 = time.time()
    gm = st['all_img'][..., 3:]
    loss = 100 * torch.nn.functional.mse_loss(b['shaded'][..., 3:], gm) + OI.image_loss(b['shaded'][..., :3] * gm, st['all_img'][..., :3] * gm, 'l1', 'log_srgb') + \
        (1 - OI.ssim((b['shaded'][..., :3] * gm).permute(0, 3, 1, 2), (st['all_img'][..., :3] * gm).permute(0, 3, 1, 2))) + \
        OI.sdf_reg_loss(sdf, edges) * 0.2 + e + b['msdf_image'].abs().mean() + \
        torch.nn.functional.mse_loss(torch.nn.functional.normalize(b['geometric_normal'][..., :3], dim=-1), st['all_normal'])
    T['losses'] = time.time() - t0
    t0 = time.time()
    loss.backward()
    T['backward'] = time.time() - t0
    # scale every stage to the full workload by its own size law
    npt_s, npt_f = verts.shape[0], (grid_n_full + 1) ** 3
    tet_s, tet_f = tets.shape[0], 6 * grid_n_full ** 3
    px_s, px_f = res_s * res_s, res_full * res_full * frames_full
    surf = (grid_n_full / n_s) ** 2                                    # surface vertices / faces grow with the square of the grid resolution
    full = {'sdf_sweep_fwd': T['sdf_sweep_fwd'] * npt_f / npt_s, 'marching_tets': T['marching_tets'] * tet_f / tet_s,
            'lbs': T['lbs'] * surf * frames_full, 'render': T['render'] * (px_f / px_s), 'eikonal_fwd': T['eikonal_fwd'] * samples_full / eik_s,
            'losses': T['losses'] * px_f / px_s}
    fwd_s = sum(T[k] for k in full)
    full['backward'] = T['backward'] * sum(full.values()) / max(fwd_s, 1e-9)       # the backward scales like the forward it differentiates
    total = sum(full.values())
    return {'value': 1.0 / total, 'unit': 'iters/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': (f'oracle chain (oracle/tick.py stages: SDF sweep, marching tets, LBS, render with the numpy/torch rasteriser, eikonal, losses, '
                       f'autograd backward) on Kuhn n={n_s} ({npt_s} pts / {tet_s} tets), 1 frame {res_s}x{res_s}, {eik_s} eikonal samples: '
                       f'{sum(T.values()):.1f} s of CPU work; every stage scaled to n={grid_n_full}, {frames_full} x {res_full}^2, {samples_full} samples by '
                       f'its own size law (points, tets, surface ~ n^2, pixels, samples)'),
            'stage_seconds_sample': {k: round(v, 4) for k, v in T.items()}, 'stage_seconds_full_scaled': {k: round(v, 3) for k, v in full.items()}}


def parity_summary(rep):
    """the whole-tick parity report (oracle/parity.py) as scalars for the short line: triangle indices bit-equal, the worst relative loss
    difference and the worst max-norm gradient difference with the per-pixel winners shared, and the count of discrete raster differences"""
    sh = rep.get('shared_raster') or {}
    gd = [v for k, v in (sh.get('max_rel_grad_diff') or {}).items() if v is not None and k != 'sdf_net_bias']      # (the cancelling bias sums: reported in the detail file)
    out = {'faces_bit_equal': bool(rep.get('mesh_faces_equal')), 'max_rel_loss_diff': sh.get('max_rel_loss_diff'),
           'max_rel_grad_diff': max(gd) if gd else None, 'raster_ids_differ': rep.get('raster_ids_differ'),
           'alpha_pixels_differ': rep.get('alpha_pixels_differ')}
    f64 = rep.get('float64')
    if f64:         # worst relative-L2 gradient error of the GPU tick and of the fp32 oracle tick against the float64 evaluation of the oracle chain
        mx = lambda d: max([v for v in d.values() if v is not None] or [None])
        out['gpu_vs_float64_l2'], out['oracle32_vs_float64_l2'] = mx(f64['gpu_l2']), mx(f64['oracle32_l2'])
    return out


def cpu_baseline_config2(sc2):
    """BASELINE configs[1] -- 1 frame, tet-res 64 (Kuhn n = 32), 512 x 512, mask loss -- through the WHOLE oracle tick (oracle/tick.py:
    tick_init, the pinned CPU restatement of the reference's tick_init: SDF sweep, marching tets, SMPL-X LBS, render with the CPU
    rasteriser, mask loss, the eikonal term on 50 000 samples, sdf_reg) + autograd backward of the config's total, timed in full on the
    host cores: no extrapolation.  The state is the config-2 GPU scene's (`sc2`, after its timed steps), copied to the host; the optimiser
    step (~1e-3 of the tick) is not included.
    The very same oracle run is the checker of a FULL-SIZE parity comparison (oracle/parity.py): one GPU tick of `sc2` on the same
    parameters, target, background, surface samples and shading jitter, every loss term and every parameter gradient compared
    (`parity` in the returned dict; outside any timed region of the GPU measurement)."""
    import torch
    from oracle import parity as OP
    rep, tm = OP.scene_tick_parity(sc2, iteration=10, seed=0, truth64=True)
    fwd, bwd = tm['forward_s'], tm['backward_s']
    return {'value': 1.0 / (fwd + bwd), 'unit': 'iters/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': (f'BASELINE configs[1] in full, no extrapolation: the whole oracle tick (oracle/tick.py:tick_init -- SDF sweep over '
                       f'{sc2.geometry.verts.shape[0]} grid vertices, marching tets over {sc2.geometry.indices.shape[0]} tets, LBS, 512x512 render with the '
                       f'numpy/torch rasteriser, mask loss, eikonal term on 50000 samples, sdf_reg) {fwd:.1f} s + autograd backward of the mask loss '
                       f'{bwd:.1f} s on {torch.get_num_threads()} host threads; mesh {rep["mesh_verts"]} vertices / {rep["mesh_faces"]} faces'),
            'config': 'configs[1]: 1 frame, tet-res 64, 512x512, mask loss only', 'forward_s': fwd, 'backward_s': bwd, 'parity': rep,
            'parity_summary': parity_summary(rep)}


def timed_steps(step, k, warm=3):
    """ms per step of `step()` over k steps after `warm` untimed ones (one process, stream-synchronised on both sides)"""
    import gc
    import torch
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    gc.collect()
    t0 = time.time()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    return (time.time() - t0) / k * 1e3


def virtual_rank_ms(sc, W, r, mode, steps, eik_total=50000):
    """Step time (ms) of scene `sc` run as rank r of a W-rank job on THIS GPU (d3h.dist_ops virtual-rank mode): mode 'shard' = the job's
    default (1/W of the grid sweep + its backward, eik_total / W eikonal samples, gradient bucket), 'replicate' = every rank repeats the
    frame-independent work.  The learning rates are set to zero first (and stay there): every kernel of the step still runs, the
    parameters -- hence the other ranks' sweep shards, refreshed here -- stay exact.  The collectives' wire time is NOT in the number."""
    from d3h import dist_ops as D
    F = sc.FLAGS
    saved = (sc.world, sc.rank, getattr(F, 'sdf_shard', None), getattr(F, 'eikonal_samples', eik_total))
    sc.world, sc.rank = W, r
    D.set_virtual(r, W)
    F.sdf_shard, F.eikonal_samples = None, eik_total
    sc.freeze_learning()
    if mode == 'shard':
        sc.enable_work_sharding(eik_total)
        sc.refresh_virtual()
    step = {'split': sc.step_split, 'seq': sc.step_seq}.get(sc.loss_set, sc.step)
    try:
        ms = timed_steps(step, steps)
    finally:
        D.set_virtual()
        sc.world, sc.rank, F.sdf_shard, F.eikonal_samples = saved
    return ms


def predicted_scaling(sc_frames, t1_weak_ms, t1_strong_ms, n_grid, bucket_bytes, steps, frames_weak=4, frames_strong=8):
    """The table DESIGN.md section 4 quotes: for W in {2, 4, 8}, the measured step time of ONE virtual rank (virtual_rank_ms) for the
    sharded default and for full replication, plus the modelled wire time of the step's collectives (d3h.dist_ops.model_collective_us, both
    a single-link ring and the direct all-links form), and what follows for the two ways BASELINE counts: weak (frames_weak frames per GPU;
    frames/s relative to one GPU) and strong (frames_strong frames in total, BASELINE configs[3]; optimiser steps/s relative to one GPU
    running all of them).  sc_frames: {frames per rank: Scene}; the scenes are left with zero learning rates."""
    from d3h import dist_ops as D
    out = {'model': {'link_GBps_one_direction': D.XGMI_LINK_GBPS, 'efficiency': D.XGMI_EFFICIENCY, 'latency_us_per_collective': D.MEASURED_FLOOR_US or D.COLLECTIVE_LATENCY_US,
                     'note': 'rank_ms measured on one GPU in virtual-rank mode (no wire time); collective_ms modelled, fully exposed (no overlap assumed); '
                             'predicted = rank_ms + collective_ms(ring over one link, the pessimistic form)'},
           'one_gpu_ms': {'weak_%d_frames' % frames_weak: t1_weak_ms, 'strong_%d_frames' % frames_strong: t1_strong_ms}, 'weak': {}, 'strong': {}}
    for W in (2, 4, 8):
        r = W // 2
        for kind, frames, t1 in (('weak', frames_weak, t1_weak_ms), ('strong', frames_strong // W, t1_strong_ms)):
            sc = sc_frames.get(frames)
            if sc is None:
                continue
            row = {'frames_per_rank': frames, 'rank_index': r}
            for mode in ('shard', 'replicate'):
                ms = virtual_rank_ms(sc, W, r, mode, steps)
                coll = [('all_reduce', bucket_bytes)] + ([('all_gather', 4 * n_grid), ('reduce_scatter', 4 * n_grid)] if mode == 'shard' else [])
                ring = sum(D.model_collective_us(k, b, W, links=1) for k, b in coll) / 1e3
                direct = sum(D.model_collective_us(k, b, W, links=W - 1) for k, b in coll) / 1e3
                pred = ms + ring
                e = {'rank_ms': ms, 'collective_ms_model': {'ring_one_link': ring, 'direct_all_links': direct}, 'predicted_ms': pred}
                if kind == 'weak':
                    e['frames_per_s'] = W * frames / pred * 1e3
                    e['weak_eff'] = t1 / pred                              # (frames/s at W) / (W x frames/s at 1)
                else:
                    e['optimizer_steps_per_s'] = 1e3 / pred
                    e['strong_x'] = t1 / pred
                row[mode] = e
            # shard or replicate for this W, from the numbers (not a default): the measured rank times + the modelled collectives
            row['chosen'] = 'shard' if row['shard']['predicted_ms'] <= row['replicate']['predicted_ms'] else 'replicate'
            out[kind][str(W)] = row
    return out


def launch_ranks(n, argv):
    """--gpus N > 1 without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <argv>` as a child
    process (this parent has not touched the GPU and never will), relay its stdout, return its exit code"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.lstrip().startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0:
        sys.stderr.write(f'bench.py: the {n}-rank job exited with code {rc}; no result line\n')
        return rc
    if line is None:
        sys.stderr.write(f'bench.py: the {n}-rank job printed no result line\n')
        return 1
    got = json.loads(line).get('n_gpus')
    if got != n:
        sys.stderr.write(f'bench.py: the job reports n_gpus = {got}, asked for {n}\n')
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='3', help="BASELINE.json config (1-based): 2 = res64/512^2/1 frame/mask, 3 = res128/1024^2/4 frames/full (the metric), 5 = split stage; 6 = seq stage (extra); "
                                                  "f3c = the reference's own working point (configs/f3c.json: batch 1, 1080x1080, tet grid 128, init-stage loss stack with the MobileNetV2 normal loss)")
    ap.add_argument('--frames-total', type=int, default=0, help='strong scaling (BASELINE configs[3]): this many frames in total, split over the ranks')
    ap.add_argument('--shard-sweep', action='store_true', help='(accepted for compatibility: sharding the frame-independent work is the default for N > 1)')
    ap.add_argument('--both-modes', action='store_true', help='N > 1: measure the other mode (sharded <-> replicated) too, even with --no-extras (the default full run does)')
    ap.add_argument('--replicate', action='store_true', help='N > 1: replicate the SDF sweep and all eikonal samples on every rank (one collective per step)')
    ap.add_argument('--as-rank-of', type=int, default=0, help='one GPU stands in for one rank of a W-rank job (virtual-rank mode); prints that rank\'s step time')
    ap.add_argument('--rank-index', type=int, default=-1, help='which rank --as-rank-of plays (default W // 2: the slice of the grid with the most surface)')
    ap.add_argument('--no-predict', action='store_true', help='skip the predicted 2/4/8-GPU table of the N = 1 run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the 12-buffer rate')
    ap.add_argument('--all-buffers', action='store_true', help='time the step with all 12 reference buffers composited + antialiased (render.py:430-449) instead of the loss-consumed three')
    ap.add_argument('--prefit', type=int, default=300)
    args = ap.parse_args()
    args.config = int(args.config) if str(args.config).isdigit() else str(args.config)

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # parent: no GPU call before or after
    claim_stdout()                     # (before any library that may print to stdout is loaded)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and os.environ.get('D3H_MIOPEN_SHARED_CACHE') != '1':
        # Hygiene: every rank of a node JIT-compiles the same MIOpen kernels on a cold cache (the MobileNetV2 / AlexNet trunks of --config f3c / 5);
        # one user cache directory per rank keeps them from writing the same files.  (It does NOT cure the cold two-process-one-GPU loopback fault of
        # round 4's "open observation": 2 of 8 cold runs died with per-rank caches, 1 of 8 with the shared one -- profiles/r5_hazard_summary.md section 2.)
        base = os.path.join(os.environ.get('TMPDIR', '/tmp'), f'd3h_miopen_{os.getuid()}', f'rank{rank}')
        os.makedirs(base, exist_ok=True)
        os.environ.setdefault('MIOPEN_USER_DB_PATH', base)
        os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', base)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to report a mislabelled run')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the product has no CPU path)')
    if args.as_rank_of and (world != 1 or args.as_rank_of < 2):
        raise SystemExit('--as-rank-of W (W >= 2) runs on ONE GPU without a process group: use it with --gpus 1')
    if world > 1 and os.environ.get('D3H_SHARE_GPU') != '1' and torch.cuda.device_count() < world:
        raise SystemExit(f'bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible')
    if os.environ.get('D3H_SHARE_GPU') == '1':
        local = 0            # plumbing check of the N > 1 path on a one-GPU box: every rank on cuda:0, D3H_DIST_BACKEND=gloo (RCCL refuses that)
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(backend=os.environ.get('D3H_DIST_BACKEND', 'nccl'), init_method='env://')         # "nccl" is RCCL on ROCm
    dev = f'cuda:{local}'

    from d3h import scene, sdf_mlp, _lib as L
    if args.config == 2:
        cfg = dict(res=512, grid_n=32, n_frames=1, loss_set='mask')
        name = 'config2: 1 frame, tet-res 64 (Kuhn n=32: 35937 verts / 196608 tets), 512x512, mask loss only'
    elif args.config == 5:
        cfg = dict(res=1024, grid_n=63, n_frames=4, loss_set='split')
        name = ('config5 (per GPU): dual garment+body pass (hmSDF_Tets cloth + body, tick_split x2 per iteration), tet-res 128, 1024x1024, '
                '4 frames; mSDF partitioned as after the split stage has converged (garment = the torso band, body = the rest: both passes extract '
                'a mesh of several thousand faces, see config.split_faces); loss stack of tick_split with the MSE+cos normal term (no pretrained '
                'perceptual weights offline)')
    elif args.config == 'f3c':
        # /root/reference/configs/f3c.json:5-19 -- "train_res": [1080, 1080], "batch": 1, "gshell_grid": 128 -- with train.py's init stage:
        # total = reg + normal + msk (train.py:718), normal = 50 x MobileNetV2-feature L1 (hmsdf.py:895-902; seeded random trunk offline)
        cfg = dict(res=1080, grid_n=63, n_frames=1, loss_set='init')
        name = ("f3c (the reference's working point, configs/f3c.json): 1 frame, 1080x1080 (not a multiple of any kernel tile), tet-res 128, "
                "init-stage stack: mask + MobileNetV2-feature normal loss (random-init trunk: no pretrained weights offline) + sdf_reg + eikonal")
    elif args.config == 6:
        cfg = dict(res=1024, grid_n=63, n_frames=1, loss_set='seq')
        name = ('seq stage (not a BASELINE config; SURVEY 8(f) rank 1): fixed-topology body + garment mesh, MLP_deform offsets, LBS, '
                'render_mask, tick_seq loss stack, 1024x1024, 1 frame')
    else:
        cfg = dict(res=1024, grid_n=63, n_frames=4, loss_set='full')
        name = 'config3: 4-frame batch, tet-res 128 (Kuhn n=63: 262144 verts / 1500282 tets), 1024x1024, mask+normal+SSIM+sdf_reg+eikonal'
    strong = args.frames_total > 0
    shard = world > 1 and not args.replicate             # frame-independent work split over the ranks (d3h/dist_ops.py)
    EIK_TOTAL = 50000                                    # hmsdf.py:714
    w_job = args.as_rank_of or world                     # ranks of the job this process is (or stands in for) a member of
    eik_per_rank = -(-EIK_TOTAL // w_job) if (shard or (args.as_rank_of and not args.replicate)) else EIK_TOTAL
    if strong:
        if args.frames_total % w_job:
            raise SystemExit(f'--frames-total {args.frames_total} is not divisible by {w_job} ranks')
        cfg['n_frames'] = args.frames_total // w_job
        name = (f'config4: {args.frames_total} frames frame-parallel over {w_job} GPU(s) ({cfg["n_frames"]} per GPU), tet-res 128, 1024x1024, '
                f'mask+normal+SSIM+sdf_reg+eikonal ({eik_per_rank} eikonal samples per GPU)')
    if args.config == 'f3c' and os.environ.get('D3H_MIOPEN_FIND', '1') != '0':
        torch.backends.cudnn.benchmark = True            # MIOpen's search mode for the MobileNetV2 convolutions at 1080 x 1080 (as config 5)
    lp = None
    if args.config == 5:
        # full loss stack incl. LPIPS: AlexNet trunk with a seeded random initialisation (no ImageNet weights offline) and the calibrated
        # linear layers of the reference's vendored package, kept as data in tests/golden/lpips.npz
        import numpy as np
        import lpips
        # MIOpen's search ("find") mode for the AlexNet convolutions at 4 x 1024^2: +4 s at start-up, -11 % per step against the immediate-mode
        # heuristics (31.7 vs 35.7 ms on one box, tools: see DESIGN.md section 5); D3H_MIOPEN_FIND=0 keeps the heuristics
        if os.environ.get('D3H_MIOPEN_FIND', '1') != '0':
            torch.backends.cudnn.benchmark = True
        lp = lpips.LPIPS(net='alex', pretrained=False)
        gpath = os.path.join(ROOT, 'tests', 'golden', 'lpips.npz')
        if os.path.exists(gpath):
            g = np.load(gpath)
            lp.load_state_dict({f'lin{k}.model.1.weight': torch.from_numpy(g[f'alex.lin{k}']) for k in range(5)}, strict=False)
        name += '; + LPIPS (alex trunk, random init; vendored linear layers)'
    sc = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, dist_world=world, dist_rank=rank, lpips=lp,
                     frame_seed=1234 + rank * cfg['n_frames'],
                     flags_hook=lambda F: (setattr(F, 'eikonal_samples', EIK_TOTAL), setattr(F, 'use_perceptual_normal_loss', args.config == 'f3c')), **cfg)
    if world > 1:      # identical shared parameters on every rank
        for p in sc.shared_params:
            dist.broadcast(p.data, src=0)
        if shard:
            # each rank sweeps 1/N of the tet grid (sdf all-gathered, d(sdf) reduce-scattered) and draws 50 000 / N eikonal samples
            sc.enable_work_sharding(EIK_TOTAL)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.as_rank_of:
        # ---- one GPU stands in for one rank of a W-rank job: that rank's step time, nothing else ------------------------------------------
        W = args.as_rank_of
        r = args.rank_index if args.rank_index >= 0 else W // 2
        mode = 'replicate' if args.replicate else 'shard'
        for _ in range(args.warmup):
            {'split': sc.step_split, 'seq': sc.step_seq}.get(cfg['loss_set'], sc.step)()
        ms = virtual_rank_ms(sc, W, r, mode, args.steps)
        from d3h import dist_ops as D
        nb = 4 * sum(p.numel() for p in sc._bucket_members())
        n_grid = sc.geometry.verts.shape[0]
        coll = [('all_reduce', nb)] + ([('all_gather', 4 * n_grid), ('reduce_scatter', 4 * n_grid)] if mode == 'shard' else [])
        emit({'metric': 'step time of ONE rank of a W-rank job, measured on one GPU (virtual-rank mode, no wire time)', 'value': ms, 'unit': 'ms',
                          'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': False, 'scaling': None,
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': name, 'as_rank_of': W, 'rank_index': r, 'mode': mode, 'frames_per_rank': cfg['n_frames'],
                                     'eikonal_samples_per_rank': eik_per_rank,
                                     'bucket_bytes': nb,
                                     'collective_ms_model': {'ring_one_link': sum(D.model_collective_us(k, b, W, 1) for k, b in coll) / 1e3,
                                                             'direct_all_links': sum(D.model_collective_us(k, b, W, W - 1) for k, b in coll) / 1e3}}})
        return

    if args.all_buffers:
        sc.FLAGS.render_buffers = 'all'
        name += '; ALL 12 buffers rendered'
    step = {'split': sc.step_split, 'seq': sc.step_seq}.get(cfg['loss_set'], sc.step)
    for _ in range(args.warmup):
        step()
    sync()
    # Python's cyclic GC runs ONE full (generation-2) collection over the start-up heap (~2.7e5 tracked objects: modules, the scene)
    # around iteration 25-35 of a fresh process -- 85-100 ms with the GIL held (tools/gpu_stall_hunt.py), i.e. +4 ms/step on a 20-step
    # window if it happens to fall inside.  Collect now and move the survivors to the permanent generation; the collector stays enabled.
    import gc

    def settle_heap():
        gc.collect()
        gc.freeze()
        if os.environ.get('D3H_BENCH_GC_THRESHOLD'):          # experiment: how much do the young-generation collections of the step cost?
            gc.set_threshold(int(os.environ['D3H_BENCH_GC_THRESHOLD']), 50, 50)
    settle_heap()
    lib = L.lib()
    lib.d3h_timing_read.restype = ctypes.c_int64
    if os.environ.get('D3H_BENCH_NO_KTIME') != '1':
        if os.environ.get('D3H_BENCH_NO_RESERVE') != '1':
            lib.d3h_timing_reserve(ctypes.c_int64(64 * args.steps))          # the events are created here, not inside the timed region
        # HIP events around the instrumented kernels, on their launch streams (csrc/timing.hip).  Inside the timed region only the SDF kernels
        # (ids 0-6: the `roofline` kernel and the other MFMA-bound launches, ~10 event pairs per step); the ~45 pairs of the image-space kernels
        # cost 0.12 ms per step of host time on the launch path (7.41 -> 7.28 ms), so they are timed in a second pass right after the region
        lib.d3h_timing_select(ctypes.c_uint64(SDF_KT_MASK))
        lib.d3h_timing_enable(1)
    sc.coll_timing = [] if world > 1 else None
    marks = []
    t0 = time.time()
    for _ in range(args.steps):
        marks.append(time.perf_counter())        # host clock at every step entry (no synchronisation): tells a uniformly slow run from one stall
        step()
    sync()
    dt = time.time() - t0
    marks.append(time.perf_counter())
    lib.d3h_timing_enable(0)
    recs = collect_kernel_timing(lib) if rank == 0 else []
    coll = sc.coll_timing
    sc.coll_timing = None
    if os.environ.get('D3H_BENCH_NO_KTIME') != '1':
        lib.d3h_timing_select(ctypes.c_uint64(~SDF_KT_MASK & 0xFFFFFFFFFFFFFFFF))
        lib.d3h_timing_enable(1)
        for _ in range(min(20, args.steps)):          # every rank runs them: the step contains the gradient all-reduce
            step()
        sync()
        lib.d3h_timing_enable(0)
        if rank == 0:
            recs = recs + collect_kernel_timing(lib)
        lib.d3h_timing_select(ctypes.c_uint64(0xFFFFFFFFFFFFFFFF))
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- extras outside the timed region (rank 0 reports): active-tile count of the sparse backward, coverage, the 12-buffer rate ----
    extras = {}
    md = sc.geometry.last_mesh_dict
    if 'imesh' not in md:
        md = {'imesh': md['all_mesh']}
    cov_px = None
    if 'buffers' in sc.geometry.last_mesh_dict:
        cov_px = float((sc.geometry.last_mesh_dict['buffers']['shaded'][..., 3] > 0).sum())
    dt12 = None
    if not args.no_extras and not args.all_buffers and cfg['loss_set'] == 'full':
        save = sc.FLAGS.render_buffers
        sc.FLAGS.render_buffers = 'all'                   # all 12 buffers composited + antialiased, as the reference does every iteration
        for _ in range(3):
            step()
        sync()
        settle_heap()
        k12 = max(5, args.steps // 4)
        t1 = time.time()
        for _ in range(k12):
            step()
        sync()
        dt12 = (time.time() - t1) / k12
        sc.FLAGS.render_buffers = save
    # ---- the same step with the exact-f32 MFMA kernels (D3H_SDF_X3=0), same process, same scene: what the bf16 x 3 arithmetic buys --------
    dt_f32 = None
    from d3h import sdf_mlp as _smx
    if not args.no_extras and world == 1 and getattr(_smx, 'X3', False):
        _smx.X3 = False
        try:
            dt_f32 = timed_steps(step, max(5, args.steps // 4)) * 1e-3
        finally:
            _smx.X3 = True
        for _ in range(2):
            step()                                          # (back on the default packs before anything else is measured)
        sync()
    # ---- the GPU rate of BASELINE configs[1] (the configuration the CPU baseline is timed on), single-GPU run only ----------------------
    gpu_cfg2 = sc2 = None
    if not args.no_cpu_baseline and world == 1 and args.config == 3:
        sc2 = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
        for _ in range(10):
            sc2.step()
        sync()
        settle_heap()
        t1 = time.time()
        for _ in range(60):
            sc2.step()
        sync()
        gpu_cfg2 = 60 / (time.time() - t1)
    # ---- BASELINE configs[3] in the same invocation: 8 frames in total over the ranks (8 / N per GPU), "strong" -------------------------
    cfg4 = None
    if not args.no_extras and not strong and not args.all_buffers and cfg['loss_set'] == 'full' and 8 % world == 0:
        f4 = 8 // world
        sc4 = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, dist_world=world, dist_rank=rank,
                          frame_seed=1234 + rank * f4, flags_hook=lambda F: setattr(F, 'eikonal_samples', EIK_TOTAL),
                          **dict(cfg, n_frames=f4))
        if world > 1:
            for p in sc4.shared_params:
                dist.broadcast(p.data, src=0)
            if shard:
                sc4.enable_work_sharding(EIK_TOTAL)
        for _ in range(3):
            sc4.step()
        sync()
        settle_heap()
        k4 = max(10, args.steps // 4)
        sc4.coll_timing = [] if world > 1 else None
        t1 = time.time()
        for _ in range(k4):
            sc4.step()
        sync()
        dt4 = time.time() - t1
        if world > 1:
            t = torch.tensor([dt4], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt4 = float(t.item())
        cfg4 = {'workload': f'config4: 8 frames frame-parallel over {world} GPU(s) ({f4} per GPU), tet-res 128, 1024x1024, '
                            f'mask+normal+SSIM+sdf_reg+eikonal ({eik_per_rank} eikonal samples per GPU)',
                'value': k4 / dt4, 'unit': 'iters/s', 'scaling': 'strong', 'steps': k4, 'ms_per_step': dt4 / k4 * 1e3, 'frames_per_gpu': f4,
                'frames_per_s': 8 * k4 / dt4}
        if world > 1 and sc4.coll_timing:
            us4 = [a.elapsed_time(b) * 1e3 for a, b in sc4.coll_timing]
            cfg4['collective_avg_us'] = sum(us4) / len(us4)
        del sc4
    # ---- N > 1: the OTHER mode of the frame-independent work (sharded <-> replicated) on the same scene, so that one run shows both rates
    # (the headline `value` is the mode the command line chose) -------------------------------------------------------------
    other_mode = ranks_seen = None
    if world > 1:                                            # (every rank is still here: the ranks other than 0 leave before the line is built)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    if world > 1 and ((not args.no_extras and cfg['loss_set'] == 'full') or args.both_modes):
        (sc.disable_work_sharding if shard else sc.enable_work_sharding)(EIK_TOTAL)
        for _ in range(3):
            step()
        sync()
        ko = max(10, args.steps // 4)
        t1 = time.time()
        for _ in range(ko):
            step()
        sync()
        dto = time.time() - t1
        t = torch.tensor([dto], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dto = float(t.item())
        other_mode = {'mode': 'replicate' if shard else 'shard', 'value': (1.0 if strong else world) * ko / dto, 'unit': 'iters/s', 'steps': ko,
                      'ms_per_step': dto / ko * 1e3, 'collectives_per_step': 1 if shard else 3}
        (sc.enable_work_sharding if shard else sc.disable_work_sharding)(EIK_TOTAL)
    # ---- the per-call floor of the step's three collectives at their real sizes, on RCCL: the real group for N > 1, a ONE-rank group on a
    # single-GPU run (launch + kernel floor, no wire) -- replaces the assumed 30 us in the model of predicted_scaling ----
    rccl_floor = None
    if not args.no_extras and cfg['loss_set'] in ('full', 'init', 'mask'):
        from d3h import dist_ops as _D
        made = False
        try:
            if world == 1 and not dist.is_initialized():
                import socket
                with socket.socket() as so:
                    so.bind(('127.0.0.1', 0))
                    port = so.getsockname()[1]
                dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
                made = True
            fl = _D.measure_rccl_floor_us(sc.geometry.verts.shape[0], 4 * sum(p.numel() for p in sc._bucket_members()), dev)
            rccl_floor = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in fl.items() if k != 'bytes'}
            rccl_floor['bytes'] = fl['bytes']
        except Exception as e:          # measurement only: never fail the run
            sys.stderr.write(f'bench.py: RCCL floor measurement skipped: {e!r}\n')
        finally:
            if made:
                dist.destroy_process_group()
    # ---- predicted 2 / 4 / 8-GPU table: one virtual rank of each world size, measured here, + the xGMI model of the collectives ----------
    last_loss = {k: float(v) for k, v in sc.last.items()}
    predicted = None
    if world == 1 and not args.no_extras and not args.no_predict and not strong and not args.all_buffers and cfg['loss_set'] == 'full' and cfg4 is not None:
        scs = {cfg['n_frames']: sc}
        for f_ in (2, 1):
            scs[f_] = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, frame_seed=1234,
                                  flags_hook=lambda F: setattr(F, 'eikonal_samples', EIK_TOTAL), **dict(cfg, n_frames=f_))
            for _ in range(5):
                scs[f_].step()
        predicted = predicted_scaling(scs, dt / args.steps * 1e3, cfg4['ms_per_step'], sc.geometry.verts.shape[0],
                                      4 * sum(p.numel() for p in sc._bucket_members()), max(20, args.steps // 4), frames_weak=cfg['n_frames'])
        for f_ in (2, 1):
            del scs[f_]
    if world > 1:
        dist.barrier()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rooflines: per kernel id, grouped by launch size; algorithmic work / mean event time ----
    # the SDF kernels are grouped by launch size (the grid sweep and the 50 000-sample sweeps are different workloads of one kernel); every
    # other kernel by id, with the mean work per launch (mesh-dependent sizes differ from iteration to iteration)
    by, tot_units = {}, {}
    for kid, units, ms in recs:
        key = (kid, units) if (kid <= 6 or kid == 27) else (kid, -1)
        by.setdefault(key, []).append(ms)
        tot_units[key] = tot_units.get(key, 0) + units
    n_grid = sc.geometry.verts.shape[0] if not getattr(sc.FLAGS, 'sdf_shard', None) else None
    frames = cfg['n_frames']
    npix = frames * cfg['res'] * cfg['res']
    rooflines = []
    main_roof = None
    from d3h import sdf_mlp as _sm
    x3_on = bool(getattr(_sm, 'X3', False))
    h2_on = x3_on and bool(getattr(_sm, 'H2', False))
    H2_NOTE = ('fp32 GEMMs on the fp16 matrix pipe: every operand = two fp16 numbers (value + residual x 2^11), three v_mfma_f32_16x16x32_f16 per '
               '16x16x32 block, fp32 accumulate (csrc/sdf_mlp_x3.h "h2": error vs float64 equal to plain fp32 evaluation); achieved = algorithmic '
               'fp32 FLOP/s, peak = fp16 dense peak 2500 / 3')
    X3_NOTE = ('fp32 GEMMs on the bf16 matrix pipe: every operand = three bf16 numbers, six v_mfma_f32_16x16x32_bf16 per 16x16x32 block, fp32 '
               'accumulate (csrc/sdf_mlp_x3.h: error vs float64 no larger than the exact-f32 MFMA path); achieved = algorithmic fp32 FLOP/s, '
               'peak = bf16 dense peak 2500 / 6')
    for (kid, units), v in sorted(by.items()):
        nm, bound, work, unit, note = KT[kid]
        avg = sum(v) / len(v)
        if units < 0:
            units = tot_units[(kid, units)] / len(v)
        eff_units = units
        if unit == 'covered pixel':
            # the renders of one step have different coverage (loss render / watertight render); the last loss render's count is used
            eff_units = cov_px if cov_px else units
        if unit == 'active point':
            eff_units = None                  # the active count lives on the device; reported as time only
        if kid == 9:
            work_total = 16.0 * units + 72.0 * (cov_px or 0)
        elif kid in (11, 14, 24):
            work_total = 8.0 * units + 16.0 * npix            # + the raster read
        elif kid == 16:
            work_total = 4.0 * units + (28.0 + 24.0) * npix   # + references in, SSIM planes out
        elif eff_units is None or work is None:
            work_total = None
        else:
            work_total = float(work) * eff_units
        is_h2 = h2_on and kid in h2_kernel_ids(_sm)
        if x3_on and kid in X3_KERNEL_IDS:          # the bf16 x 3 twin of the kernel (csrc/sdf_mlp_x3.hip, the *_x3_kernel templates of sdf_mlp_bwd.hip)
            nm = nm.replace('sdf_mlp_fwd_kernel', 'sdf_mlp_fwd_x3_kernel<.., NP = 2 (fp16 x 2)>' if is_h2 else 'sdf_mlp_fwd_x3_kernel') \
                   .replace('sdf_mlp_bwd_data_kernel', 'sdf_mlp_bwd_data_x3_kernel<.., NP = 2 (fp16 x 2)>' if is_h2 else 'sdf_mlp_bwd_data_x3_kernel') \
                   .replace('sdf_mlp_bwd_dw_layers_kernel', 'sdf_mlp_bwd_dw_layers_h2_kernel' if is_h2 else 'sdf_mlp_bwd_dw_layers_x3_kernel')
        e = {'kernel': nm, 'bound': bound, 'launch_ms': avg, 'launches': len(v), 'units_per_launch': int(units), 'unit': unit, 'note': note}
        if kid in LIMITER:
            # latency / atomic-rate bound: a fraction of the HBM roof would say nothing; the algorithmic byte rate stays for reference
            e['bound'], e['limiter'] = LIMITER[kid]
            e.update({'peak': None, 'frac': None, 'unit_rate': 'GB/s (algorithmic, informational)'})
            if work_total is not None:
                e['achieved'] = work_total / (avg * 1e-3) / 1e9
        elif work_total is not None:
            if bound == 'mfma':
                e.update({'achieved': work_total / (avg * 1e-3) / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit_rate': 'TFLOP/s'})
                if x3_on and kid in X3_KERNEL_IDS:
                    prods = H2_PRODUCTS if is_h2 else X3_PRODUCTS
                    e.update({'peak': MFMA_BF16_PEAK_TFLOPS / prods, 'arithmetic': H2_NOTE if is_h2 else X3_NOTE, 'products_per_fp32_product': prods,
                              'executed_mfma_tflops': e['achieved'] * prods, 'exact_f32_mfma_peak': MFMA_F32_PEAK_TFLOPS})
            else:
                e.update({'achieved': work_total / (avg * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS, 'unit_rate': 'GB/s'})
            e['frac'] = e['achieved'] / e['peak']
            eik_cus = getattr(sc.geometry, '_eik_cus', 256)
            if kid <= 3 and e.get('frac') is not None and eik_cus < 256 and units == int(getattr(sc.FLAGS, 'eikonal_samples', 50000)):
                # launched on a subset of the CUs on purpose (geometry/hmsdf.py:_eikonal_async): the rest run the other stream's kernels
                e['cus'] = eik_cus
                e['frac_of_cus_used'] = e['frac'] * 256.0 / eik_cus
        rooflines.append(e)
        if kid == 0 and (main_roof is None or units > main_roof['units_per_launch']):
            main_roof = e
    if main_roof is not None:
        n_pts = main_roof['units_per_launch']
        nosave = x3_on and bool(getattr(_sm, 'RECOMPUTE', False))          # the training sweep writes no activations (the backward recomputes what it visits)
        kname = 'sdf_mlp_fwd_kernel<false, %d>'
        if x3_on:
            kname = 'sdf_mlp_fwd_x3_kernel<false, %d, ' + ('false' if nosave else 'true') + (', 2>' if h2_on else ', 3>')
        roof = {'kernel': kname % (0 if (n_pts + 127) // 128 >= 1024 else 1),
                'bound': 'mfma', 'achieved': main_roof['achieved'],
                'peak': main_roof['peak'], 'unit': 'TFLOP/s', 'frac': main_roof['frac'],
                'traffic': ((PMC_TRAFFIC_BYTES_H2_NOSAVE if h2_on else PMC_TRAFFIC_BYTES_X3_NOSAVE) if nosave else (PMC_TRAFFIC_BYTES_X3 if x3_on else PMC_TRAFFIC_BYTES)).get(n_pts),
                'traffic_note': 'bytes/launch from rocprofv3 PMC (profiles/, FETCH_SIZE x2 + WRITE_SIZE)' + ('' if nosave else ', incl. 1.88 GB saved activations'),
                'launch_ms': main_roof['launch_ms'], 'launches': main_roof['launches'], 'points_per_launch': int(n_pts),
                'algorithmic_GBps': BYTES_PER_POINT_FWD * n_pts / (main_roof['launch_ms'] * 1e-3) / 1e9}
        if x3_on:
            prods = H2_PRODUCTS if h2_on else X3_PRODUCTS
            roof.update({'arithmetic': H2_NOTE if h2_on else X3_NOTE, 'products_per_fp32_product': prods, 'executed_mfma_tflops': main_roof['achieved'] * prods,
                         'exact_f32_mfma_peak': MFMA_F32_PEAK_TFLOPS, 'frac_of_exact_f32_mfma_peak': main_roof['achieved'] / MFMA_F32_PEAK_TFLOPS,
                         # the round-5 kernel (bf16 x 3, six products) was priced against 2500 / 6: the same algorithmic rate against THAT roof, for comparison
                         'frac_of_bf16x3_roof': main_roof['achieved'] / (MFMA_BF16_PEAK_TFLOPS / X3_PRODUCTS)})
    else:
        roof = None
    if strong:
        value, scaling = args.steps / dt, 'strong'
    else:
        value, scaling = world * args.steps / dt, 'weak'
    out = {'metric': 'train iters/sec @ tet-res 128, 1024^2 render; 1/2/4/8 MI355X', 'value': value, 'unit': 'iters/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
           'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'dtype_note': ('fp32 storage, fp32 accumulation, fp32-level results throughout; the SDF network\'s forward / tangent / data-backward GEMMs are '
                          'evaluated on the matrix cores from split operands (csrc/sdf_mlp_x3.h): ' + ('three fp16 products of two-plane fp16 splits (gradient-valued operands scaled by a power of two per launch); D3H_SDF_H2=0 = ' if h2_on else '') + 'six bf16 products of three-way bf16 splits; D3H_SDF_X3=0 = exact-f32 MFMA') if x3_on else 'fp32 throughout (exact-f32 MFMA)',
           'config': {'workload': name, 'frames_per_gpu': cfg['n_frames'], 'mesh_verts': int(md['imesh'].v_pos.shape[0]),
                      'mesh_faces': int(md['imesh'].t_pos_idx.shape[0]),
                      'watertight_render': "FLAGS.visualize_watertight = True (train.py:1627); inside tick_* the watertight twin is not rendered (no loss reads it and a tick returns loss values only) -- render_* called directly and the 'all' mode of all_12_buffers_iters_per_s render it",
                      'buffers': "what tick_init reads (shaded, geometric_normal, msdf_image): the default of tick_* through the unmodified train.py; FLAGS.render_buffers = 'all' gives all_12_buffers_iters_per_s",
                      'parallelism': f'frame-parallel dp{world}' + (' + SDF sweep and eikonal samples sharded over the ranks' if shard else
                                                                     (' (frame-independent work replicated on every rank)' if world > 1 else '')),
                      'optimizer': 'one-launch fused Adam (d3h.optim.FusedAdam)' if sc.opt is not None else 'torch.optim.Adam(fused=True) x2',
                      'covered_pixels_last_render': cov_px, 'loss': last_loss,
                      'step_entry_intervals_ms': (lambda d: {'p50': d[len(d) // 2], 'p99': d[min(len(d) - 1, int(len(d) * 0.99))], 'max': d[-1],
                                                             'note': 'host clock between consecutive step entries inside the timed region (the host '
                                                                     'runs ahead of the GPU by design); a max far above p50 = one stall, e.g. host jitter'})(
                          sorted(1e3 * (b - a) for a, b in zip(marks[:-1], marks[1:])))},
           'roofline': roof, 'rooflines': rooflines,
           'rooflines_note': ('HIP events on the launch streams: the SDF-network kernels (the first entries, incl. `roofline`) inside the timed '
                              'region, the image-space / mesh kernels in a second pass of %d steps right after it (their ~45 event pairs per '
                              'step cost 0.12 ms of host time on the launch path)' % min(20, args.steps))}
    if getattr(sc, 'split_faces', None):
        out['config']['split_faces'] = sc.split_faces
    if lp is not None:
        # the LPIPS trunk (AlexNet features through MIOpen) as a roofline entry of its own: the prediction side of one tick_split call, forward +
        # input gradient, timed in isolation with HIP events; x 2 calls per iteration.  FLOPs: the five convolutions at this resolution (2 x MACs)
        r_ = cfg['res']
        o1 = (r_ + 4 - 11) // 4 + 1
        o2 = (o1 - 3) // 2 + 1
        o3 = (o2 - 3) // 2 + 1
        conv_flop = 2.0 * (o1 * o1 * 64 * 3 * 121 + o2 * o2 * 192 * 64 * 25 + o3 * o3 * (384 * 192 * 9 + 256 * 384 * 9 + 256 * 256 * 9))
        x_ = torch.rand(cfg['n_frames'], 3, r_, r_, device=dev, requires_grad=True)
        ref_ = lp.reference_features(torch.rand(cfg['n_frames'], 3, r_, r_, device=dev))

        def trunk():
            x_.grad = None
            lp(x_, None, ref_features=ref_).mean().backward()
        for _ in range(3):
            trunk()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            trunk()
        e1.record()
        torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1) / 10
        fl_ = 2 * conv_flop * cfg['n_frames']                      # forward + data gradient (the trunk is frozen: no weight gradients)
        out['config']['lpips_trunk'] = {'kernel': 'AlexNet features of LPIPS through MIOpen (5 convolutions + ReLU / max-pool / bias / head), forward + input gradient of ONE tick_split call',
                                        'bound': 'mfma', 'ms_per_call': ms_, 'calls_per_iteration': 2, 'conv_gflop_per_call': fl_ / 1e9,
                                        'achieved': fl_ / (ms_ * 1e-3) / 1e12, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit_rate': 'TFLOP/s',
                                        'frac': fl_ / (ms_ * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                        'note': 'library kernels (MIOpen find mode), timed in isolation; share of the step = 2 x ms_per_call / ms_per_step'}
    if dt12 is not None:
        out['config']['all_12_buffers_iters_per_s'] = (1.0 if strong else world) / dt12
    if dt_f32 is not None:
        out['config']['exact_f32_mfma_iters_per_s'] = 1.0 / dt_f32       # D3H_SDF_X3=0: the SDF GEMMs on v_mfma_f32_16x16x4_f32 / 32x32x2_f32
    if cfg4 is not None:
        out['config']['config4_frames_total_8'] = cfg4
    if predicted is not None:
        out['config']['predicted_scaling'] = predicted
    if rccl_floor is not None:
        out['config']['rccl_floor_us'] = rccl_floor
    if world > 1:
        out['config']['world_size'] = dist.get_world_size()               # what RCCL sees
        out['config']['backend'] = dist.get_backend()
        out['config']['rccl_ranks_seen'] = ranks_seen                     # an all-reduce of ones: the ranks that took part in a collective
        out['config']['mode'] = 'shard' if shard else 'replicate'
        if other_mode is not None:
            out['config']['other_mode'] = other_mode
    if world > 1 and coll:
        us = [a.elapsed_time(b) * 1e3 for a, b in coll]
        out['config']['collective'] = {'kind': 'all_reduce(sum) of one flat fp32 gradient bucket per step', 'bytes': int(getattr(sc, 'bucket_bytes', 0)),
                                       'avg_us': sum(us) / len(us), 'calls': len(us),
                                       'extra_collectives_per_step': 2 if shard else 0,
                                       'extra_collectives': ('all_gather of the sdf shards + reduce_scatter of d(sdf), %d bytes each' % (4 * sc.geometry.verts.shape[0]))
                                       if shard else None}
        out['optimizer_steps_per_s'] = args.steps / dt
        out['frames_per_s'] = world * cfg['n_frames'] * args.steps / dt
        out['value_note'] = ('value = N x K / T counts rank-iterations (the bench contract: units all ranks processed / time); one OPTIMISER step of '
                             'the job consumes N x %d frames and takes T / K: optimizer_steps_per_s and frames_per_s are the training-speed figures'
                             % cfg['n_frames'])
    if not args.no_cpu_baseline and world == 1:          # the CPU baseline is measured once, on the single-GPU run
        # measured: BASELINE configs[1] through the whole oracle tick; beside it the GPU rate of the SAME config, and -- as a second, stated
        # ESTIMATE -- the config-3 figure assembled from per-stage timings scaled by each stage's size law
        # the state the oracle tick is timed (and compared) on: the REPRODUCIBLE one of tests/test_gpu_fullsize.py -- the CPU-fitted SDF network
        # of tests/golden/parity_state_sdf.npz + seeded host-side deform / trans fields (no GPU optimiser step: every box compares the same scene)
        fixture = os.path.join(ROOT, 'tests', 'golden', 'parity_state_sdf.npz')
        if os.path.exists(fixture):
            scp = scene.Scene(device=dev, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask', sdf_state=fixture)
            scp.perturb_state_seeded(0)
        else:
            scp = sc2
            if scp is None:
                scp = scene.Scene(device=dev, prefit_steps=args.prefit, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
                for _ in range(10):
                    scp.step()
        cb = cpu_baseline_config2(scp)
        if gpu_cfg2 is not None:
            cb['gpu_same_config_iters_per_s'] = gpu_cfg2
        cb['config3_extrapolated'] = cpu_baseline_config3_scaled(cfg['grid_n'], cfg['res'], cfg['n_frames'])
        out['cpu_baseline'] = cb
    emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
