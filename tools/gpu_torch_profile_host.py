"""torch.profiler, HOST side: where the CPU time of one training iteration goes (main thread AND the autograd thread) -- aten ops and
autograd nodes by self / total CPU time.  VIRT=W FRAMES=F: the step of one virtual rank of a W-rank job.
    VIRT=8 FRAMES=1 python tools/gpu_torch_profile_host.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from d3h.scene import Scene

VIRT = int(os.environ.get('VIRT', 0))
sc = Scene(res=int(os.environ.get('RES', 1024)), grid_n=int(os.environ.get('GRID', 63)), n_frames=int(os.environ.get('FRAMES', 4)), device='cuda',
           prefit_steps=300, loss_set=os.environ.get('LOSS', 'full'), visualize_watertight=True)
for _ in range(8):
    sc.step()
if VIRT:
    from d3h import dist_ops as D
    sc.world, sc.rank = VIRT, VIRT // 2
    D.set_virtual(sc.rank, VIRT)
    sc.freeze_learning()
    sc.enable_work_sharding(50000)
    sc.refresh_virtual()
    for _ in range(5):
        sc.step()
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        sc.step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
print('---- by SELF cpu time, us per iteration ----')
for e in rows[:45]:
    print(f'{e.self_cpu_time_total / N:9.1f} self  {e.cpu_time_total / N:9.1f} total  n/iter {e.count / N:6.1f}  {e.key[:90]}')
rows = sorted(ka, key=lambda e: -e.cpu_time_total)
print('---- by TOTAL cpu time, us per iteration ----')
for e in rows[:45]:
    print(f'{e.cpu_time_total / N:9.1f} total  {e.self_cpu_time_total / N:9.1f} self  n/iter {e.count / N:6.1f}  {e.key[:90]}')
