"""x3 (bf16 triple-split) forward sweep vs the exact-f32 MFMA sweep on the GPU: agreement and time.   python tools/gpu_probe_x3.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'd3human-code_amd'), os.path.join(ROOT, 'tests')]
import torch
from d3h import sdf_mlp
from parity_cases import golden, sd_from_golden, T
dev = 'cuda'
g = golden('sdf_mlp.npz'); sd = sd_from_golden(g, dev)
wp = sdf_mlp.pack_weights(sd); wp3 = sdf_mlp.pack_weights3(sd)
x = T(g['x'], dev); ref = g['sdf'].reshape(-1)
o1, a1, _ = sdf_mlp.forward(x, wp, save=True)
o3, a3, _ = sdf_mlp.forward(x, wp, save=True, wp3=wp3)
print('n', x.shape[0], 'f32 vs golden', np.abs(o1.cpu().numpy() - ref).max(), 'x3 vs golden', np.abs(o3.cpu().numpy() - ref).max(),
      'x3 vs f32', (o3 - o1).abs().max().item(), 'act diff', (a3 - a1).abs().max().item(), 'act scale', a1.abs().max().item())
torch.manual_seed(0)
for n in (262144, 50000, 32768, 6250):
    xs = (torch.rand(n, 3, device=dev) * 2 - 1)
    for name, kw in (('f32', {}), ('x3', {'wp3': wp3})):
        for save in (True, False):
            for _ in range(3):
                sdf_mlp.forward(xs, wp, save=save, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                sdf_mlp.forward(xs, wp, save=save, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            print(f'n {n:7d} {name:4s} save={save!s:5s} {dt * 1e6:8.1f} us   {n * 826880 / dt / 1e12:7.1f} TFLOP/s (fp32-equivalent)', flush=True)
    a = sdf_mlp.forward(xs, wp); b = sdf_mlp.forward(xs, wp, wp3=wp3)
    print('   random points: x3 vs f32 max abs', (a - b).abs().max().item(), 'sign flips', int(((a > 0) != (b > 0)).sum()))
