"""micro-benchmark: antialias forward / backward at the bench image size on a real rasterisation"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT)
import torch
from d3h import raster, mtets, synth
C = int(sys.argv[1]) if len(sys.argv) > 1 else 9
v, t = (torch.from_numpy(a) for a in synth.kuhn_grid(63))
sdf = synth.body_sdf(v)
o = mtets.marching_tets(v.cuda(), sdf.cuda(), torch.ones(v.shape[0]).cuda(), t.cuda())
verts, tri = o['verts'], o['faces32']
mv, mvp, campos = synth.camera(1024)
B = 4
offs = torch.tensor([[0.02 * b, 0.0, 0.0] for b in range(B)]).cuda()
vh = torch.cat([verts[None] + offs[:, None], torch.ones(B, verts.shape[0], 1).cuda()], -1)
clip = (vh @ torch.from_numpy(mvp).cuda().T).contiguous()
rast, db = raster.rasterize(clip, tri, (1024, 1024))
print('faces', tri.shape[0], 'coverage', float((rast[..., 3] > 0).float().mean()))
col = torch.rand(B, 1024, 1024, C, device='cuda').requires_grad_(True)
pos = clip.clone().requires_grad_(True)
G = torch.randn(B, 1024, 1024, C, device='cuda')
def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
out = raster.antialias(col, rast, pos, tri)
fwd = t_ms(lambda: raster.antialias(col.detach(), rast, pos.detach(), tri))
def fb():
    o = raster.antialias(col, rast, pos, tri)
    o.backward(G)
    col.grad = None; pos.grad = None
both = t_ms(fb)
cp = t_ms(lambda: col.detach().clone())
mb = col.numel() * 4 / 1e6
print(f'C={C}: image {mb:.0f} MB; aa fwd {fwd*1e3:.0f} us ({(2*mb + 67)/fwd/1e3:.2f} TB/s on 2x image + rast); fwd+bwd {both*1e3:.0f} us; torch clone {cp*1e3:.0f} us ({2*mb/cp/1e3:.2f} TB/s)')
