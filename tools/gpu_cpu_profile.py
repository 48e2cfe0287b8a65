"""Host-side profile of a training iteration (cProfile over Scene.step* on the GPU box): where the Python / launch time goes and where
the host blocks on the GPU (`tolist`, `.cpu()`, `nonzero`, `synchronize` entries are stream synchronisations).
    python tools/gpu_cpu_profile.py [full|split|seq]
    VIRT=W FRAMES=F python tools/gpu_cpu_profile.py full      the step of one virtual rank of a W-rank job with F frames per rank"""
import cProfile, pstats, io, os, re, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene

mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
VIRT = int(os.environ.get('VIRT', 0))
cfg = dict(res=int(os.environ.get('RES', 1024)), grid_n=int(os.environ.get('GRID', 63)), n_frames=1 if mode == 'seq' else int(os.environ.get('FRAMES', 4)), device='cuda', prefit_steps=int(os.environ.get('PREFIT', 300)), loss_set=mode,
           visualize_watertight=(mode != 'seq'))
sc = Scene(**cfg)
step = {'split': sc.step_split, 'seq': sc.step_seq}.get(mode, sc.step)
for _ in range(10):
    step()
if VIRT:
    from d3h import dist_ops as D
    sc.world, sc.rank = VIRT, VIRT // 2
    D.set_virtual(sc.rank, VIRT)
    sc.freeze_learning()
    if os.environ.get('REPLICATE') != '1':
        sc.enable_work_sharding(50000)
        sc.refresh_virtual()
    for _ in range(5):
        step()
import gc
gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    step()
torch.cuda.synchronize()
print('ms/step', (time.time() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(70)
rows = []
for l in s.getvalue().split('\n'):
    m = re.match(r'\s*(\d+)(?:/\d+)?\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', l)
    if m:
        rows.append((float(m.group(2)), float(m.group(4)), int(m.group(1)), m.group(6)[-100:]))
for r in rows[:65]:
    print(f'self {r[0] * 1e3 / 20:7.3f} ms/step  cum {r[1] * 1e3 / 20:7.3f}  n/step {r[2] / 20:6.1f}  {r[3]}')
