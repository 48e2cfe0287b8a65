"""Host timeline of the stretch between the marching-tets read-back and the first heavy render kernel (config 3, no profiler): when does each
host function start / end, relative to the return of read-back #1?  gpurun -- 'python tools/dbg/gpu_host_window.py'"""
import os
import statistics as st
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=512, grid_n=32, n_frames=1, loss_set='mask') if os.environ.get('CONFIG') == '2' else \
    scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
ev = []


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    lab = label or name

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            ev.append((lab, t0, time.perf_counter()))
    setattr(obj, name, g)


import kaolin.ops.mesh as K
from d3h import mtets, raster, imgops, texmlp, lbs as dlbs
from render import render as R
from geometry import hmsdf
wrap(K, 'sample_points')
wrap(sc.geometry, '_eikonal_async')
wrap(sc.geometry, '_launch_eikonal')
wrap(sc.geometry, '_eikonal_finish')
wrap(sc.geometry.smplx_deform, 'nearest')
wrap(sc.geometry.smplx_deform, 'lbs_forward_batch')
wrap(R, 'render_mesh')
wrap(raster, 'rasterize')
wrap(raster, 'gbuffer')
wrap(imgops, 'composite')
wrap(raster, 'antialias')
wrap(texmlp, 'texture_mlp')
wrap(imgops, 'pixel_losses')
wrap(sc.geometry, '_sdf_sweep')
orig = torch.Tensor.tolist


def tl(self):
    if self.is_cuda:
        t0 = time.perf_counter()
        r = orig(self)
        ev.append(('readback', t0, time.perf_counter()))
        return r
    return orig(self)


torch.Tensor.tolist = tl
_es = torch.cuda.Event.synchronize


def es(self):
    t0 = time.perf_counter()
    r = _es(self)
    ev.append(('readback', t0, time.perf_counter()))       # speculative extraction: the sizes arrive behind an event on the copy stream
    return r


torch.cuda.Event.synchronize = es
wrap(mtets._MTetsFn, 'forward', 'mtets_forward')
wrap(mtets.TetGrid, '_wait', 'readback')          # speculative extraction: the sizes arrive in host memory, the host spins on a flag
per_step = []
for _ in range(100):
    ev.clear()
    t0 = time.perf_counter()
    sc.step()
    per_step.append((t0, list(ev), time.perf_counter()))
torch.cuda.synchronize()
torch.Tensor.tolist = orig
rows = {}
for t0, evs, t1 in per_step:
    rb = [e for e in evs if e[0] == 'readback']
    z = rb[0][2]                                         # return of read-back #1
    seen = {}
    for lab, a, b in evs:
        k = seen.get(lab, 0)
        seen[lab] = k + 1
        rows.setdefault((lab, k), []).append(((a - z) * 1e6, (b - z) * 1e6))
    rows.setdefault(('step', 0), []).append(((t0 - z) * 1e6, (t1 - z) * 1e6))
print('label                     start_us   end_us   (median over 100 steps, relative to the return of read-back #1)')
for (lab, k), v in sorted(rows.items(), key=lambda kv: st.median(x[0] for x in kv[1])):
    print(f'{lab + "#" + str(k):24s} {st.median(x[0] for x in v):9.0f} {st.median(x[1] for x in v):9.0f}')
