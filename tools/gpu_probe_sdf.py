"""GPU probe (dev tool): SDF MLP fwd parity vs the oracle + timing at tet-res-128 size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch, torch.nn as nn, numpy as np
from d3h import _lib as L
from oracle import sdf_mlp as O

torch.manual_seed(0)
dims = [(39, 256), (256, 256), (256, 256), (256, 256), (295, 256), (256, 256), (256, 256), (256, 1)]
sd = {}
for li, (i, o) in enumerate(dims):
    l = nn.Linear(i, o); sd[f'net.{2*li}.weight'] = l.weight.detach(); sd[f'net.{2*li}.bias'] = l.bias.detach()
dev = 'cuda'
g = lambda k: sd[k].to(dev).contiguous()
wh = torch.stack([sd[f'net.{i}.weight'] for i in (2, 4, 6, 10, 12)]).to(dev).contiguous()
bh = torch.stack([sd[f'net.{i}.bias'] for i in (2, 4, 6, 10, 12)]).to(dev).contiguous()
lib = L.lib()
wp = torch.zeros(lib.d3h_sdf_mlp_wpack_floats(), device=dev)
keep = [g('net.0.weight'), g('net.0.bias'), wh, bh, g('net.8.weight'), g('net.8.bias'), g('net.14.weight'), g('net.14.bias')]
L.check(lib.d3h_sdf_mlp_pack(*[L.ptr(t) for t in keep], L.ptr(wp), L.stream()), 'pack')
for n in (1, 100, 4096, 5000):
    x = torch.rand(n, 3) * 2.4 - 1.2
    ref = O.mlp_forward(x, sd).flatten()
    xd = x.to(dev); out = torch.empty(n, device=dev)
    L.check(lib.d3h_sdf_mlp_fwd(L.ptr(xd), None, L.f32(0), L.ptr(wp), L.ptr(out), None, None, L.i64(n), L.stream()), 'fwd')
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    flips = ((out.cpu() > 0) != (ref > 0)).sum().item()
    print(f'n={n} max_abs_err={err:.3e} sign_flips={flips} ref_absmax={ref.abs().max():.3e}')
n = 262144
x = (torch.rand(n, 3, device=dev) * 2.4 - 1.2); out = torch.empty(n, device=dev)
act = torch.empty(lib.d3h_sdf_mlp_act_floats(n), device=dev)
for save in (False, True):
    for _ in range(3):
        lib.d3h_sdf_mlp_fwd(L.ptr(x), None, L.f32(0), L.ptr(wp), L.ptr(out), None, L.ptr(act) if save else None, L.i64(n), L.stream())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 20
    for _ in range(K):
        lib.d3h_sdf_mlp_fwd(L.ptr(x), None, L.f32(0), L.ptr(wp), L.ptr(out), None, L.ptr(act) if save else None, L.i64(n), L.stream())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print(f'fwd n={n} save={save}: {ms:.3f} ms  {n*826880/ms/1e9:.1f} TFLOP/s  ({n*16/ms/1e6:.2f} GB/s algorithmic)')
