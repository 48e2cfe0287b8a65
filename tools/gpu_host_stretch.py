"""Host timeline of the stretch between the first marching-tets read-back and the launch of the eikonal chain (the GPU is nearly idle there:
profiles/r3_bench_config3_timeline.csv), mean over the timed iterations.   python tools/gpu_host_stretch.py [n_steps]"""
import collections, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
torch.cuda.set_device(0)
from d3h import scene, mtets, sdf_mlp
import kaolin.ops.mesh as km
from render import render as R, mesh as M
from geometry import hmsdf as H
from deform import smplx_exavatar_deformer as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ev = []
mark = lambda s: ev.append((s, time.perf_counter()))


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        mark(label + ' >')
        try:
            return f(*a, **k)
        finally:
            mark(label + ' <')
    setattr(obj, name, g)


tl = torch.Tensor.tolist
def tolist(self):
    r = tl(self)
    mark('read-back returned')
    return r
torch.Tensor.tolist = tolist
wrap(mtets._MTetsFn, 'forward', 'MTetsFn.forward')
wrap(D.SMPLX_Deformer, 'nearest', 'nearest (1-NN)')
wrap(D.SMPLX_Deformer, 'lbs_forward_batch', 'lbs_forward_batch')
wrap(mtets, 'marching_tets', 'marching_tets')
wrap(km, 'sample_points', 'sample_points')
wrap(sdf_mlp, 'eikonal_begin', 'eikonal_begin')
wrap(H.HmSDFTetsGeometry, '_launch_eikonal', '_launch_eikonal')
wrap(R, 'render_mesh', 'render_mesh')
sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, dist_world=1, dist_rank=0, lpips=None, frame_seed=1234,
                 flags_hook=lambda F: setattr(F, 'eikonal_samples', 50000), res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(10):
    sc.step()
import gc
gc.collect(); gc.freeze()
acc = collections.OrderedDict()
for _ in range(n):
    ev.clear()
    sc.step()
    # from the FIRST read-back return to the render_mesh entry
    i0 = next(i for i, (s, _) in enumerate(ev) if s == 'read-back returned')
    t0 = ev[i0][1]
    seen = collections.Counter()
    for s, t in ev[i0:]:
        seen[s] += 1
        key = s if seen[s] == 1 else f'{s} #{seen[s]}'
        a = acc.setdefault(key, [0.0, 0])
        a[0] += t - t0; a[1] += 1
        if s == 'render_mesh >':
            break
torch.cuda.synchronize()
print('host time after the first read-back returned, us (mean):')
for k, (t, c) in acc.items():
    print(f'  {t / c * 1e6:8.1f}  {k}')
