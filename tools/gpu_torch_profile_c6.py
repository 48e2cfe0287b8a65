"""torch.profiler view of one config-5 iteration (garment + body tick_split with LPIPS): which aten ops launch the library elementwise kernels."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from d3h.scene import Scene
import lpips
torch.backends.cudnn.benchmark = True
lp = lpips.LPIPS(net='alex', pretrained=False)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lpips.npz'))
lp.load_state_dict({f'lin{k}.model.1.weight': torch.from_numpy(g[f'alex.lin{k}']) for k in range(5)}, strict=False)
sc = Scene(res=1024, grid_n=63, n_frames=1, device='cuda', prefit_steps=int(os.environ.get('PREFIT', 300)), loss_set="seq", visualize_watertight=True)
for _ in range(6):
    sc.step_seq()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        sc.step_seq()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
dt = lambda e: getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
rows = [e for e in ka if dt(e) > 0 and (e.key.startswith('aten::') or 'Backward' in e.key)]
rows.sort(key=lambda e: -dt(e))
print('---- ops with device time of their own (per iteration) ----')
tot = 0
for e in rows[:120]:
    tot += dt(e) / 2
    print(f'{dt(e) / 2:9.1f} us  n {e.count / 2:6.1f}  {e.key:34s} {str(e.input_shapes)[:120]}')
print('sum', tot)
