"""torch.profiler view of one training iteration on the GPU box: which autograd nodes / aten ops own the library elementwise kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'd3human-code_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from d3h.scene import Scene

sc = Scene(res=512, grid_n=32, n_frames=1, device='cuda', prefit_steps=int(os.environ.get('PREFIT', 300)), loss_set='mask', visualize_watertight=True)
for _ in range(8):
    sc.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3):
        sc.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
# the same by call site: the innermost frames of this build in the Python stack of each aten op that launched device work
ks = prof.key_averages(group_by_input_shape=True, group_by_stack_n=12)
srows = [e for e in ks if e.key.startswith('aten::') and getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)) > 0]
srows.sort(key=lambda e: -getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0)))
print('---- aten ops with device time of their own, by call site (3 iterations) ----')
for e in srows[:70]:
    own = [f.split('d3human-code_amd/')[-1] for f in (e.stack or []) if 'd3human-code_amd' in f][:3]
    t = getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
    print(f'{t / 3:8.1f} us/iter  n/iter {e.count / 3:5.1f}  {e.key:26s} {str(e.input_shapes)[:70]:70s} {" <- ".join(own)[:150]}')
rows = [e for e in ka if e.key.startswith('aten::')]
tot = lambda e: getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0))
rows.sort(key=lambda e: -tot(e))
print('---- aten ops by device time (3 iterations) ----')
for e in rows[:40]:
    print(f'{tot(e) / 3:9.1f} us/iter  n/iter {e.count / 3:6.1f}  {e.key:28s} {str(e.input_shapes)[:110]}')
print(ka.table(sort_by='cuda_time_total', row_limit=45, max_name_column_width=60, max_shapes_column_width=70))
