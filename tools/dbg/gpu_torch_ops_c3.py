"""Which aten ops are left in a config-3 step (the torch glue around the library's kernels)?  torch.profiler over 20 steps, per-op launch counts
and device time per step, with the python call site of the big ones.  gpurun -- 'python tools/dbg/gpu_torch_ops_c3.py'"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
N = 20
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N):
        sc.step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if e.key.startswith('aten::') and dt > 0:
        rows.append((dt / N, e.count / N, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print('us/step  calls/step  op  (call site)')
for r in rows[:45]:
    print(f'{r[0]:7.1f} {r[1]:6.1f}  {r[2]:28s} {r[3]}')
print('total aten device us/step', round(sum(r[0] for r in rows), 1), 'calls/step', round(sum(r[1] for r in rows), 1))
