"""Does a side-stream eikonal chain disturb the main stream's sweep backward?  The sweep backward's dx is deterministic (no atomics): it is
recomputed many times while another stream runs eikonal_loss chains, and compared bitwise.   python tools/dbg/gpu_dbg_x3_concurrent.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'd3human-code_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from d3h import sdf_mlp
from parity_cases import golden
g = golden('sdf_mlp.npz')
keys = sdf_mlp._PARAM_ORDER
ps = [torch.from_numpy(g['sd.net.' + k]).cuda().requires_grad_(True) for k in keys]
torch.manual_seed(0)
n = int(os.environ.get('N', 2197))
x = (torch.rand(n, 3, device='cuda') * 2 - 1).requires_grad_(True)
go = torch.zeros(n, 1, device='cuda')
sel = torch.randperm(n, device='cuda')[: n // 6]
go[sel] = torch.randn(sel.shape[0], 1, device='cuda')
pts = torch.rand(50000, 3, device='cuda') * 1.6 - 0.8
side = torch.cuda.Stream()
pk = sdf_mlp.PackedWeights(ps)

def sweep_bwd():
    x.grad = None
    y = sdf_mlp.sdf_query(x, ps, pack=pk)
    (y * go).sum().backward()
    return x.grad.clone(), ps[4].grad.clone()

ref, refw = sweep_bwd()
for p in ps: p.grad = None
torch.cuda.synchronize()
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if os.environ.get('SIDE', '1') == '1':
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e = sdf_mlp.eikonal_loss(pts, ps, 0.05, pack=pk)
    dx, dw = sweep_bwd()
    for p in ps: p.grad = None
    torch.cuda.synchronize()
    if not torch.equal(dx, ref):
        bad += 1
        d = (dx - ref).abs().sum(dim=-1)
        nz = torch.nonzero(d > 0).reshape(-1)
        print(it, 'dx differs on', nz.numel(), 'points, tiles', sorted(set((nz // 16).tolist()))[:12], 'max', float(d.max()), 'dw rel', float((dw - refw).norm() / refw.norm()), flush=True)
print('bad', bad)
