import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
import lpips
from geometry.perceptual import MobileNetPerceptualLoss
def t_ms(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for net in ('alex', 'vgg'):
        m = lpips.LPIPS(net=net, pretrained=False).cuda()
        for chl in (False, True):
            a = torch.rand(4, 3, 1024, 1024, device='cuda', requires_grad=True); b = torch.rand(4, 3, 1024, 1024, device='cuda')
            if chl:
                m = m.to(memory_format=torch.channels_last)
            def f():
                x = a.contiguous(memory_format=torch.channels_last) if chl else a
                y = b.contiguous(memory_format=torch.channels_last) if chl else b
                m(x, y).mean().backward(); a.grad = None
            print(f'benchmark={bench} {net} channels_last={chl}: fwd+bwd {t_ms(f):.2f} ms', flush=True)
    mn = MobileNetPerceptualLoss(use_gpu=True)
    a = torch.rand(1, 3, 1024, 1024, device='cuda', requires_grad=True); b = torch.rand(1, 3, 1024, 1024, device='cuda')
    def g():
        mn(a, b).backward(); a.grad = None
    print(f'benchmark={bench} mobilenet 1x1024^2 fwd+bwd {t_ms(g):.2f} ms', flush=True)
