"""GPU probe (dev tool): SDF MLP fwd/bwd timing at tet-res-128 size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch, torch.nn as nn, numpy as np
from d3h import _lib as L, sdf_mlp
torch.manual_seed(0)
dev = 'cuda'
dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
params = []
for i, o in dims:
    l = nn.Linear(i, o); params += [l.weight.detach().to(dev).requires_grad_(True), l.bias.detach().to(dev).requires_grad_(True)]
n = 262144
x = (torch.rand(n, 3, device=dev) * 2.4 - 1.2)
deform = torch.zeros(n, 3, device=dev, requires_grad=True)
def timeit(fn, K=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K
sd = {k: p for k, p in zip(sdf_mlp._PARAM_ORDER, params)}
wp = sdf_mlp.pack_weights(sd, prefix='')
ms = timeit(lambda: sdf_mlp.forward(x, wp))
print(f'fwd kernel only: {ms:.3f} ms {n*826880/ms/1e9:.1f} TFLOP/s')
def fb():
    y = sdf_mlp.sdf_query(x, params, deform=deform, disp=0.003)
    y.sum().backward()
ms = timeit(fb)
print(f'fwd+bwd (autograd, incl. pack): {ms:.3f} ms  {n*826880*3/ms/1e9:.1f} TFLOP/s')
