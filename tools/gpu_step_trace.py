"""per-iteration duration of the config-3 step (event per iteration on the main stream, no host sync inside), bench.py's exact scene
    python tools/gpu_step_trace.py [n_steps]   (from a checkout root)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
torch.cuda.set_device(0)
from d3h import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 130
sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, dist_world=1, dist_rank=0, lpips=None, frame_seed=1234,
                 flags_hook=lambda F: setattr(F, 'eikonal_samples', 50000), res=1024, grid_n=63, n_frames=4, loss_set='full')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
info = []
torch.cuda.synchronize()
ev[0].record()
t0 = time.time()
for i in range(n):
    sc.step()
    ev[i + 1].record()
    info.append(sc.geometry.last_mesh_dict['imesh'].v_pos.shape[0])
torch.cuda.synchronize()
print('wall ms/step over all', (time.time() - t0) / n * 1e3)
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a in range(0, n, 10):
    seg = ms[a:a + 10]
    print(f'it {a:4d}-{a + len(seg) - 1:4d}: mean {sum(seg) / len(seg):7.3f} ms  max {max(seg):7.3f}  verts {info[a]}  ' + ' '.join(f'{x:.2f}' for x in seg))
# ---- second pass: who stalls? Python's cyclic GC (gc.callbacks) vs the caching allocator (device-level hipMalloc count), per iteration
import gc
gcev = []
def cb(phase, info):
    if phase == 'start':
        gcev.append([time.time(), info['generation'], None])
    else:
        gcev[-1][2] = time.time() - gcev[-1][0]
gc.callbacks.append(cb)
sc2 = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, dist_world=1, dist_rank=0, lpips=None, frame_seed=1234,
                  flags_hook=lambda F: setattr(F, 'eikonal_samples', 50000), res=1024, grid_n=63, n_frames=4, loss_set='full')
torch.cuda.synchronize()
for i in range(60):
    a = torch.cuda.memory_stats().get('num_device_alloc', 0)
    g0 = len(gcev)
    t0 = time.time()
    sc2.step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) * 1e3
    b = torch.cuda.memory_stats().get('num_device_alloc', 0)
    if dt > 12:
        print(f'slow iteration {i}: {dt:.1f} ms; device allocs {b - a}; gc runs', [(g, f'{(d or 0) * 1e3:.1f} ms') for _, g, d in gcev[g0:]])
print('gc gen2 runs:', [(f'{d * 1e3:.1f} ms') for _, g, d in gcev if g == 2 and d])
