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
N = int(os.environ.get('STEPS', 100))
gc.collect(); gc.freeze()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(N):
    step()
torch.cuda.synchronize()
print('ms/step', (time.time() - t0) / N * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt, ct, nc, f'{fn[-60:]}:{line}({name})', callers))
rows.sort(key=lambda r: -r[0])
print(f'total self time {sum(r[0] for r in rows) * 1e3 / N:.3f} ms/step (under cProfile)')
for r in rows[:int(os.environ.get('TOP', 70))]:
    print(f'self {r[0] * 1e3 / N:7.4f} ms/step  cum {r[1] * 1e3 / N:7.4f}  n/step {r[2] / N:6.1f}  {r[3]}')
if os.environ.get('CALLERS', '1') == '1':
    # who calls the allocation / fill / copy builtins (each is a tiny launch or an allocator round trip)
    print('--- callers of the torch builtins')
    for r in rows:
        if r[3].startswith('~:0(') and r[2] / N >= 1 and any(k in r[3] for k in ('zeros', 'zero_', 'empty', 'fill_', 'copy_', 'cat', 'clone', 'add', 'mul', 'contiguous', 'full', 'ones', 'stack', 'sum', 'to', 'float', 'tensor')):
            print(f'{r[3]}  n/step {r[2] / N:.1f}  self {r[0] * 1e3 / N:.4f}')
            for (cfn, cl, cn), v in sorted(r[4].items(), key=lambda kv: -kv[1][0] if isinstance(kv[1], tuple) else -kv[1]):
                n = v[0] if isinstance(v, tuple) else v
                print(f'      {n / N:6.1f}  {cfn[-50:]}:{cl}({cn})')
