"""GPU: the stand-alone SSIM forward + backward at 1024^2 for 3 / 6 / 12 / 24 / 48 planes: a duration that does not follow the plane count is a
dependent-chain (latency) bound, one that does is a throughput bound"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
from d3h import imgops
for planes in (3, 6, 12, 24, 48):
    B = planes // 3
    a = torch.rand(B, 3, 1024, 1024, device='cuda').requires_grad_(True)
    b = torch.rand(B, 3, 1024, 1024, device='cuda')
    def run():
        a.grad = None
        imgops.ssim(a, b).backward()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print('planes %3d  fwd+bwd %.1f us' % (planes, e0.elapsed_time(e1) * 1e3 / 20))
