"""Host-side profile of the training iteration (cProfile over Scene.step on the GPU box): where the Python / launch time goes."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene

cfg = dict(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=int(os.environ.get('PREFIT', 100)), loss_set='full', visualize_watertight=True)
sc = Scene(**cfg)
for _ in range(10):
    sc.step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
print('ms/step', (time.time() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('cumulative')
ps.print_stats(70)
print(s.getvalue()[:14000])
