"""torch.profiler view of one training iteration on the GPU box: which autograd nodes / aten ops own the library elementwise kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from d3h.scene import Scene

sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=int(os.environ.get('PREFIT', 300)), loss_set='full', visualize_watertight=True)
for _ in range(8):
    sc.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        sc.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.key.startswith('aten::')]
tot = lambda e: getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0))
rows.sort(key=lambda e: -tot(e))
print('---- aten ops by device time (3 iterations) ----')
for e in rows[:40]:
    print(f'{tot(e) / 3:9.1f} us/iter  n/iter {e.count / 3:6.1f}  {e.key:28s} {str(e.input_shapes)[:110]}')
print(ka.table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60, max_shapes_column_width=70))
