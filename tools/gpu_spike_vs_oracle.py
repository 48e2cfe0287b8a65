"""GPU tool: does the ORACLE chain produce the same gradient spike as the HIP path on the same batch?

Runs the synthetic init-stage fit at reduced size, watches max |d total / d sdf_net| per iteration, and on the first iterations whose
gradient exceeds `thresh` x the running median snapshots the state BEFORE the optimiser step (parameters, background, eikonal
samples), evaluates the oracle chain (oracle/tick.py, CPU autograd) on the snapshot and prints both gradients side by side, per
tensor and per loss term.  Usage: python tools/gpu_spike_vs_oracle.py [iters] [res] [grid_n] [frames] [max_spikes]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from d3h.scene import Scene
import e2e_cases as E
from oracle import tick as OTK

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
grid_n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
max_spikes = int(sys.argv[5]) if len(sys.argv) > 5 else 2
sc = Scene(res=res, grid_n=grid_n, n_frames=frames, device='cuda', prefit_steps=300, loss_set='full', body_verts=4096)
sc.FLAGS.eikonal_samples = 20000
import kaolin
orig_sample = kaolin.ops.mesh.sample_points
last_pts = {}
def rec(v, f, n, *a, **k):
    o = orig_sample(v, f, n, *a, **k)
    last_pts['p'] = o[0][0].detach().clone()
    return o
kaolin.ops.mesh.sample_points = rec
g, F = sc.geometry, sc.FLAGS
hist, spikes = [], 0
for it in range(iters):
    bg = torch.rand(sc.n_frames, sc.res, sc.res, 3, device=sc.device)
    tgt = sc.target(bg)
    sc._zero_grad()
    r = g.tick_init(sc.glctx, tgt, None, sc.material, sc.loss_fn, sc.it, None)
    total = r['d3h_total']
    total.backward()
    gr = E.scene_grads(sc)
    gmax = max(float(v.abs().max()) for k, v in gr.items() if k.startswith('sd.') and v is not None)
    med = float(np.median(hist)) if len(hist) >= 5 else None
    nv = g.last_mesh_dict['imesh'].v_pos.shape[0]
    print(f'it {it} total {float(total):.4f} msk {float(r["msk_loss"]):.3f} nrm {float(r["normal_loss"]):.4f} verts {nv} gmax(sdf) {gmax:.3e}', flush=True)
    if med is not None and gmax > 1e3 * med and spikes < max_spikes:
        spikes += 1
        print(f'==== spike at it {it}: gmax {gmax:.3e} vs median {med:.3e}; evaluating the oracle chain on the snapshot ...', flush=True)
        st = E.state_from_scene(sc, bg, last_pts.get('p'), sc.it)
        t0 = time.time()
        ro = OTK.tick_init(st, buffers=('shaded', 'geometric_normal', 'msdf_image'), keep=True)
        for k in ('geometric_normal', 'shaded', 'msdf_image'):
            ro['_buffers'][k].retain_grad()
        ro['_mesh']['posed'].retain_grad(); ro['_mesh']['verts'].retain_grad(); ro['_mesh']['sdf'].retain_grad()
        ro['total'].backward()
        print(f'oracle tick: {time.time() - t0:.1f} s; losses (hip | oracle):', flush=True)
        for k in ('msk_loss', 'img_loss', 'normal_loss', 'ssim_loss', 'eik_loss', 'sdf_reg_loss'):
            print(f'   {k:14s} {float(r[k]):.6f} | {float(ro[k]):.6f}')
        og = E.oracle_grads(st)
        print('   max|grad| per tensor (hip | oracle | max abs diff):')
        for k in og:
            a, b = gr[k], og[k]
            if a is None or b is None:
                continue
            a = a.detach().cpu()
            print(f'   {k:18s} {float(a.abs().max()):.3e} | {float(b.abs().max()):.3e} | {float((a - b).abs().max()):.3e}')
        bn = ro['_buffers']['geometric_normal'].grad
        print(f'   oracle: max|d total/d geometric_normal buffer| {float(bn.abs().max()):.3e}; |d/d posed| {float(ro["_mesh"]["posed"].grad.abs().max()):.3e}; '
              f'|d/d canonical verts| {float(ro["_mesh"]["verts"].grad.abs().max()):.3e}; |d/d sdf| {float(ro["_mesh"]["sdf"].grad.abs().max()):.3e}')
        # which term: oracle per-term gradient w.r.t. trans
        st2 = E.state_from_scene(sc, bg, last_pts.get('p'), sc.it)
        ro2 = OTK.tick_init(st2, buffers=('shaded', 'geometric_normal', 'msdf_image'))
        for k in ('msk_loss', 'normal_loss', 'ssim_loss'):
            gt, = torch.autograd.grad(ro2[k], st2['trans'], retain_graph=True)
            gh, = torch.autograd.grad(r[k], F.trans_optim, retain_graph=True) if False else (None,)
            print(f'   oracle d {k}/d trans max {float(gt.abs().max()):.3e}')
        torch.save({'state': {k: v for k, v in st.items() if k not in ('normal_loss_fn',)}}, os.path.join(ROOT, 'gpurun_out', f'spike_state_{it}.pt'))
    hist.append(gmax)
    sc._optimizer_step()
    sc.it += 1
print('done; spikes analysed:', spikes)
