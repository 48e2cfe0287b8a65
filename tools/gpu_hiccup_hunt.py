"""Which iterations of a bench-like run are slow, and what happened in them?  Per iteration: wall time (host clock, no sync inside), device
allocations of the caching allocator, garbage collections (gc.callbacks).  The heap is settled as bench.py does (collect + freeze).
    python tools/gpu_hiccup_hunt.py [n_steps] [repeats]"""
import gc, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
torch.cuda.set_device(0)
from d3h import scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gcev = []
def cb(phase, info):
    if phase == 'start':
        gcev.append([time.perf_counter(), info['generation'], None])
    else:
        gcev[-1][2] = time.perf_counter() - gcev[-1][0]
gc.callbacks.append(cb)
for rep in range(reps):
    sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, dist_world=1, dist_rank=0, lpips=None, frame_seed=1234,
                     flags_hook=lambda F: setattr(F, 'eikonal_samples', 50000), res=1024, grid_n=63, n_frames=4, loss_set='full')
    for _ in range(10):
        sc.step()
    torch.cuda.synchronize()
    gc.collect(); gc.freeze()
    ts, allocs, gcs = [], [], []
    torch.cuda.synchronize()
    for i in range(n):
        a0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
        g0 = len(gcev)
        t0 = time.perf_counter()
        sc.step()
        if i % 8 == 7:
            torch.cuda.synchronize()          # bounded run-ahead: the wall time of an iteration stays attributable
        ts.append(time.perf_counter() - t0)
        allocs.append(torch.cuda.memory_stats().get('num_device_alloc', 0) - a0)
        gcs.append([(g, round(d * 1e3, 2)) for _, g, d in gcev[g0:] if d is not None])
    torch.cuda.synchronize()
    tot = sum(ts)
    srt = sorted(ts)
    print(f'rep {rep}: {n} steps, mean {tot / n * 1e3:.3f} ms, median {srt[n // 2] * 1e3:.3f}, p99 {srt[int(n * 0.99)] * 1e3:.3f}, max {srt[-1] * 1e3:.3f}')
    for i, t in enumerate(ts):
        if t > 3 * srt[n // 2]:
            print(f'   it {i}: {t * 1e3:.2f} ms  device allocs {allocs[i]}  gc {gcs[i]}  mesh verts {sc.geometry.last_mesh_dict["imesh"].v_pos.shape[0]}')
    print('   gen-2 collections:', [(round(d * 1e3, 1)) for _, g, d in gcev if g == 2 and d is not None][-5:], ' device allocs total', sum(allocs))
    del sc
