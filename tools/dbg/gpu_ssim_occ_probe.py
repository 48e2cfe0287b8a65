"""GPU: the fused pixel-loss + SSIM pass with a target alpha that is zero outside a blob, occupancy cells on / off: kernel times (HIP events,
serialised) and the number of occupied cells"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
from d3h import imgops
dev = 'cuda'
B, H, W, C = 4, 1024, 1024, 9
gen = torch.Generator().manual_seed(0)
st = torch.rand(B, H, W, C, generator=gen).to(dev)
cref = torch.rand(B, H, W, 4, generator=gen)
al = torch.zeros(B, H, W)
al[:, 200:840, 380:640] = 1.0           # ~16 % of the frame
cref[..., 3] = al
cref = cref.to(dev)
nref = torch.randn(B, H, W, 3, generator=gen).to(dev)
layout = {'shaded': (0, 4), 'geometric_normal': (4, 4), 'msdf_image': (8, 1)}
MODES = {'1': (True,), '0': (False,)}.get(os.environ.get('PROBE_OCC', ''), (True, False, True, False))
for occ_on in MODES:
    imgops.SSIM_OCC = occ_on
    def run():
        x = st.clone().requires_grad_(True)
        d = imgops.pixel_losses(x, layout, cref, nref, ('l1', 'log_srgb'), want_ssim=True)
        d['vec'].sum().backward()
        return d
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        d = run()
    e1.record()
    torch.cuda.synchronize()
    print('occ', occ_on, 'fwd+bwd %.1f us' % (e0.elapsed_time(e1) * 1e3 / 20), 'ssim', float(d['ssim'].detach()))
