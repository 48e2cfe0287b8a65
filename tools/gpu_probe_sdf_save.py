"""GPU probe (dev tool): fused SDF-query forward with and without the activation save, 262 144 points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch, torch.nn as nn
from d3h import sdf_mlp
torch.manual_seed(0)
dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
params = []
for i, o in dims:
    l = nn.Linear(i, o); params += [l.weight.detach().cuda(), l.bias.detach().cuda()]
n = 262144
x = (torch.rand(n, 3, device='cuda') * 2.4 - 1.2)
wp = sdf_mlp.pack_weights({k: p for k, p in zip(sdf_mlp._PARAM_ORDER, params)}, prefix='')
def T(fn, K=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / K
for save in (False, True):
    ms = T(lambda: sdf_mlp.forward(x, wp, save=save))
    print(f'save={save}: {ms:.3f} ms  {n*826880/ms/1e9:.1f} TFLOP/s')

# the same launch as the training step issues it (deform + xdef + save), isolated by idle gaps / preceded by light kernels
import time
deform = torch.zeros(n, 3, device='cuda')
def one(pre):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    pre()
    e0.record(); sdf_mlp.forward(x, wp, deform=deform, disp=0.003, save=True, want_xdef=True); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)
junk = torch.zeros(1 << 20, device='cuda')
for name, pre in (('back-to-back', lambda: None), ('after 5 ms idle', lambda: (torch.cuda.synchronize(), time.sleep(0.005))),
                  ('after 40 tiny kernels', lambda: [junk.add_(1.0) for _ in range(40)])):
    one(pre)
    ts = [one(pre) for _ in range(8)]
    print(f'{name}: {sum(ts) / len(ts):.3f} ms (min {min(ts):.3f})')
