"""How does the wave-per-triangle rasteriser (csrc/raster.hip: raster_tris + resolve, 64-bit atomicMin z-buffer) scale with the triangle
count?  north_star names a tile-binned front end; VERDICT r3: build it only if raster_tris exceeds 0.3 ms at >= 50 k triangles.
Meshes: the capsule humanoid extracted on the Kuhn n = 63 grid (9 k faces), then 1 -> 4 subdivided up to 590 k faces; 4 frames x 1024^2,
forward + backward.
    python tools/gpu_probe_raster_tris.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import raster, mtets, synth

def timed(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3

mv, mvp, campos = synth.camera(1024)
M = torch.from_numpy(mvp).cuda()
B = 4
def subdivide(verts, tri):
    """1 -> 4 midpoint subdivision (midpoints not shared between neighbours: irrelevant to a rasteriser)"""
    a, b, c = verts[tri[:, 0].long()], verts[tri[:, 1].long()], verts[tri[:, 2].long()]
    ab, bc, ca = 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a)
    F = tri.shape[0]
    nv = torch.cat([a, b, c, ab, bc, ca])
    i = lambda k: torch.arange(F, device=verts.device) + k * F
    A, Bv, C, AB, BC, CA = i(0), i(1), i(2), i(3), i(4), i(5)
    nt = torch.cat([torch.stack([A, AB, CA], 1), torch.stack([AB, Bv, BC], 1), torch.stack([CA, BC, C], 1), torch.stack([AB, BC, CA], 1)])
    return nv.contiguous(), nt.int().contiguous()


v, t = (torch.from_numpy(a) for a in synth.kuhn_grid(63))       # (finer meshes by subdivision: the triangle count is the variable here)
o = mtets.marching_tets(v.cuda(), synth.body_sdf(v).cuda(), torch.ones(v.shape[0]).cuda(), t.cuda())
verts, tri = o['verts'], o['faces32']
for level in range(4):
    if level:
        verts, tri = subdivide(verts, tri)
    offs = torch.tensor([[0.02 * b, 0.0, 0.0] for b in range(B)]).cuda()
    clip = (torch.cat([verts[None] + offs[:, None], torch.ones(B, verts.shape[0], 1).cuda()], -1) @ M.T).contiguous()
    raster.BIN_MIN_TRIS = 1 << 30                    # wave-per-triangle kernels
    us_f = timed(lambda: raster.rasterize(clip, tri, (1024, 1024)))
    ra, _ = raster.rasterize(clip, tri, (1024, 1024))
    raster.BIN_MIN_TRIS = 1                          # tile-binned kernels (round 5)
    us_b = timed(lambda: raster.rasterize(clip, tri, (1024, 1024)))
    rb, _ = raster.rasterize(clip, tri, (1024, 1024))
    same = bool(torch.equal(ra, rb))
    raster.BIN_MIN_TRIS = 1 << 30
    c2 = clip.clone().requires_grad_(True)
    def fb():
        c2.grad = None
        r, db = raster.rasterize(c2, tri, (1024, 1024))
        (r[..., :2].sum()).backward()
    us_fb = timed(fb)
    rast, _ = raster.rasterize(clip, tri, (1024, 1024))
    cov = float((rast[..., 3] > 0).float().mean())
    print(f'subdivision level {level}  faces {tri.shape[0]:8d}  coverage {cov:.3f}  rasterize fwd: wave-per-triangle {us_f:7.1f} us, tile-binned {us_b:7.1f} us (bit-identical: {same})   fwd+bwd (wave) {us_fb:7.1f} us', flush=True)
