"""which steps of a config-3 run are slow, and does the caching allocator (new segments) or the Python GC coincide with them?"""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene
sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get('generation'), time.time())))
if os.environ.get('GC_OFF') == '1':
    for _ in range(10):
        sc.step()
    gc.collect(); gc.freeze(); gc.disable()
rows = []
for k in range(150):
    st = torch.cuda.memory_stats()
    seg0, res0 = st['segment.all.current'], st['reserved_bytes.all.current']
    n0 = len(gcs)
    torch.cuda.synchronize(); t = time.time()
    sc.step()
    torch.cuda.synchronize(); dt = (time.time() - t) * 1e3
    st = torch.cuda.memory_stats()
    rows.append((k, dt, st['segment.all.current'] - seg0, (st['reserved_bytes.all.current'] - res0) >> 20, [g[1] for g in gcs[n0:] if g[0] == 'start']))
ts = sorted(r[1] for r in rows)
print(f'median {ts[len(ts) // 2]:.2f} ms  mean {sum(ts) / len(ts):.2f} ms  max {ts[-1]:.2f} ms')
for r in rows:
    if r[1] > 1.25 * ts[len(ts) // 2] or r[2] or r[4]:
        print(f'step {r[0]:3d}: {r[1]:7.2f} ms  new segments {r[2]}  reserved +{r[3]} MiB  gc generations {r[4]}')
