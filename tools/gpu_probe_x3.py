"""x3 (bf16 triple-split) and h2 (fp16 two-plane split) forward sweeps vs the exact-f32 MFMA sweep on the GPU: agreement and time.
    python tools/gpu_probe_x3.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'd3human-code_amd'), os.path.join(ROOT, 'tests')]
import torch
from d3h import sdf_mlp
from parity_cases import golden, sd_from_golden, T
dev = 'cuda'
g = golden('sdf_mlp.npz'); sd = sd_from_golden(g, dev)
wp = sdf_mlp.pack_weights(sd); wp3 = sdf_mlp.pack_weights3(sd); wph = sdf_mlp.pack_weights_h2(sd)
x = T(g['x'], dev); ref = g['sdf'].reshape(-1)
o1, a1, _ = sdf_mlp.forward(x, wp, save=True)
o3, a3, _ = sdf_mlp.forward(x, wp, save=True, wp3=wp3)
o2, a2, _ = sdf_mlp.forward(x, wp, save=True, wp3=wph)
print('h2 vs golden', np.abs(o2.cpu().numpy() - ref).max(), 'h2 vs f32', (o2 - o1).abs().max().item(), 'act diff', (a2 - a1).abs().max().item())
print('n', x.shape[0], 'f32 vs golden', np.abs(o1.cpu().numpy() - ref).max(), 'x3 vs golden', np.abs(o3.cpu().numpy() - ref).max(),
      'x3 vs f32', (o3 - o1).abs().max().item(), 'act diff', (a3 - a1).abs().max().item(), 'act scale', a1.abs().max().item())
torch.manual_seed(0)
for n in (262144, 50000, 32768, 6250):
    xs = (torch.rand(n, 3, device=dev) * 2 - 1)
    for name, kw in (('f32', {}), ('x3', {'wp3': wp3}), ('h2', {'wp3': wph})):
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
    a = sdf_mlp.forward(xs, wp); b = sdf_mlp.forward(xs, wp, wp3=wp3); c = sdf_mlp.forward(xs, wp, wp3=wph)
    print('   random points: h2 vs f32 max abs', (a - c).abs().max().item(), 'sign flips', int(((a > 0) != (c > 0)).sum()))
    print('   random points: x3 vs f32 max abs', (a - b).abs().max().item(), 'sign flips', int(((a > 0) != (b > 0)).sum()))

# the fitted body network of the parity tests against float64 (oracle restatement on the host)
from oracle import sdf_mlp as O
gf = np.load(os.path.join(ROOT, 'tests', 'golden', 'parity_state_sdf.npz'))
sdf_ = {k: torch.from_numpy(gf[k]) for k in gf.files if k.startswith('net.')}
sdd = {k: v.to(dev) for k, v in sdf_.items()}
xs = (torch.rand(65536, 3, generator=torch.Generator().manual_seed(3)) * 2.4 - 1.2)
r64 = O.mlp_forward(xs.double(), {k: v.double() for k, v in sdf_.items()}).reshape(-1)
r32 = O.mlp_forward(xs, sdf_).reshape(-1)
wpf, wp3f, wphf = sdf_mlp.pack_weights(sdd), sdf_mlp.pack_weights3(sdd), sdf_mlp.pack_weights_h2(sdd)
for name, kw in (('f32-mfma', {}), ('x3', {'wp3': wp3f}), ('h2', {'wp3': wphf})):
    o = sdf_mlp.forward(xs.to(dev), wpf, **kw).cpu().double()
    print(f'fitted net {name:8s} vs float64: max {float((o - r64).abs().max()):.3e} mean {float((o - r64).abs().mean()):.3e} sign flips {int(((o > 0) != (r64 > 0)).sum())}'
          f'   (torch fp32 on the host: max {float((r32.double() - r64).abs().max()):.3e} mean {float((r32.double() - r64).abs().mean()):.3e})')
