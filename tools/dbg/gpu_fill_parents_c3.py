"""Who asks for the zero fills / big copies left in a config-3 step?  torch.profiler events of one step: every aten::fill_ / aten::copy_ / aten::zeros
with its chain of enclosing profiler ranges (autograd node names included).  gpurun -- 'python tools/dbg/gpu_fill_parents_c3.py'"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(10):
    sc.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for _ in range(3):
        sc.step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::fill_', 'aten::copy_', 'aten::mul', 'aten::add', 'aten::add_', 'aten::cat'):
        chain = []
        p = e.cpu_parent
        while p is not None and len(chain) < 4:
            if not p.name.startswith('aten::'):
                chain.append(p.name[:60])
            p = p.cpu_parent
        cnt[(e.name, str(e.input_shapes)[:60], ' < '.join(chain))] += 1
for (name, shp, chain), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f'{n / 3:5.1f}  {name:12s} {shp:62s} {chain}')
