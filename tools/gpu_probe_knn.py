"""GPU probe: exhaustive d3h_knn1 vs grid-accelerated d3h_knn1_grid on the bench scene's posed-mesh query (time + equality)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h import lbs as HL

dev = 'cuda'
g = torch.Generator().manual_seed(0)
u = torch.randn(10475, 3, generator=g)
tmpl = (u / u.norm(dim=1, keepdim=True) * torch.tensor([0.35, 0.85, 0.2])).to(dev).contiguous()
for nq, noise in ((43000, 0.02), (43000, 0.08), (150000, 0.03)):
    pts = (tmpl[torch.randint(0, 10475, (nq,), generator=g).to(dev)] + noise * torch.randn(nq, 3, generator=g).to(dev)).contiguous()
    t0 = time.time(); grid = HL.KnnGrid(tmpl); torch.cuda.synchronize(); tb = time.time() - t0
    for name, fn in (('exhaustive', lambda: HL.knn1(pts, tmpl)), ('grid', lambda: grid.query(pts))):
        for _ in range(3): r = fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): r = fn()
        torch.cuda.synchronize()
        print(f'nq {nq} noise {noise} {name:10s} {(time.time() - t0) / 20 * 1e6:8.1f} us   grid dims {grid.g} h {grid.h:.4f} build {tb*1e3:.1f} ms')
    assert torch.equal(HL.knn1(pts, tmpl), grid.query(pts))
