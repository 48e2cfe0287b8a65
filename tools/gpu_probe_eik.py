"""GPU probe: time of the fused eikonal term (forward + backward) vs the library-GEMM double backward at the training size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from geometry.mlp import MLP

torch.manual_seed(0)
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
x = (torch.rand(50000, 3, device='cuda') - 0.5)


def fused():
    g = net.input_gradient(x)
    (0.3 * (g.pow(2).sum(-1).sqrt() - 1).pow(2).mean()).backward()


def lib():
    v = x.detach().requires_grad_(True)
    g = torch.autograd.grad(net.forward_reference(v).sum(), v, create_graph=True)[0]
    (0.3 * (g.pow(2).sum(-1).sqrt() - 1).pow(2).mean()).backward()


for name, fn in (('fused', fused), ('library', lib)):
    for p in net.parameters():
        p.grad = None
    fn()
    ref = [p.grad.clone() for p in net.parameters() if p.grad is not None]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(name, 'ms per eikonal fwd+bwd:', (time.time() - t0) / 20 * 1e3, 'grad norm', float(sum(r.norm() ** 2 for r in ref).sqrt()))
