"""Host time per phase of the config-3 iteration: perf_counter around the Python entry points of the step (no device sync added), mean over
the timed iterations.  Shows where the host thread spends the step -- blocked in the marching-tets read-backs, issuing launches, in autograd --
next to the step time.      python tools/gpu_host_trace.py [n_steps]
VIRT=W FRAMES=F: the step of one virtual rank of a W-rank job (d3h.dist_ops virtual-rank mode) with F frames per rank."""
import collections, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
torch.cuda.set_device(0)
from d3h import scene, mtets, sdf_mlp
import kaolin.ops.mesh as km
from render import render as R
from geometry import hmsdf as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
acc = collections.OrderedDict()
stack = []


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            child = stack.pop()
            if stack:
                stack[-1] += dt
            e = acc.setdefault(label, [0.0, 0.0, 0])
            e[0] += dt; e[1] += dt - child; e[2] += 1
    setattr(obj, name, g)


G = H.HmSDFTetsGeometry
wrap(scene.Scene, '_step', 'step (all)')
wrap(scene.Scene, 'target', '  target()')
wrap(scene.Scene, '_zero_grad', '  zero_grad')
wrap(G, 'tick_init', '  tick_init')
wrap(G, '_sdf_sweep', '    sdf sweep (pack + launch)')
wrap(mtets, 'marching_tets', '    marching tets (2 read-backs inside)')
wrap(G, '_launch_eikonal', '    sampling + eikonal forward sweep')
wrap(km, 'sample_points', '      kaolin.sample_points')
wrap(R, 'render_mesh', '    render_mesh')
wrap(G, '_eikonal_finish', '    eikonal chain, remaining launches')
wrap(G, '_fused_pixel_vec', '    pixel losses (+ssim) + loss head')
wrap(torch.Tensor, 'backward', '  backward()')
wrap(scene.Scene, '_optimizer_step', '  optimizer step')

if os.environ.get('BACKWARDS', '1') == '1':
    # every autograd.Function of the package: host time of its backward (runs on autograd's device thread, inside backward())
    import importlib, pkgutil, d3h, geometry, render
    seen = set()
    for pkg in (d3h, geometry, render):
        for mi in pkgutil.iter_modules(pkg.__path__, pkg.__name__ + '.'):
            try:
                mod = importlib.import_module(mi.name)
            except Exception:
                continue
            for nm, cls in list(vars(mod).items()):
                if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls not in seen \
                        and 'backward' in vars(cls):
                    seen.add(cls)
                    f = cls.backward

                    def g(*a, _f=f, _label=f'      bwd {mi.name.split(".")[-1]}.{nm}', **k):
                        t0 = time.perf_counter()
                        try:
                            return _f(*a, **k)
                        finally:
                            e = acc.setdefault(_label, [0.0, 0.0, 0])
                            dt = time.perf_counter() - t0
                            e[0] += dt; e[1] += dt; e[2] += 1
                    cls.backward = staticmethod(g)

sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, dist_world=1, dist_rank=0, lpips=None, frame_seed=1234,
                 flags_hook=lambda F: setattr(F, 'eikonal_samples', 50000), res=1024, grid_n=63, n_frames=int(os.environ.get('FRAMES', 4)), loss_set='full')
for _ in range(10):
    sc.step()
VIRT = int(os.environ.get('VIRT', 0))
if VIRT:
    from d3h import dist_ops as D
    sc.world, sc.rank = VIRT, VIRT // 2
    D.set_virtual(sc.rank, VIRT)
    sc.freeze_learning()
    sc.enable_work_sharding(50000)
    sc.refresh_virtual()
    for _ in range(5):
        sc.step()
import gc
gc.collect(); gc.freeze()
acc.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    sc.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
print(f'step {wall:.3f} ms wall; host time per iteration, ms (total / own = total minus wrapped children):')
for k, (tot, own, c) in acc.items():
    print(f'  {k:52s} {tot / n * 1e3:7.3f} {own / n * 1e3:7.3f}   calls/iter {c / n:.1f}')
