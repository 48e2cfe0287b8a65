"""GPU probe: the SDF weight-gradient kernels (dense backward over n points + eikonal second-order pass) as a function of the
split-K block count D3H_DW_S (read once by the library).  Run once per value:  D3H_DW_S=128 python tools/gpu_probe_dw.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from geometry.mlp import MLP
from d3h import sdf_mlp

torch.manual_seed(0)
net = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256).cuda()
for n in (50000, 100000):
    x = (torch.rand(n, 3, device='cuda') * 2 - 1)
    def main_bwd():
        net.zero_grad()
        y = net(x.requires_grad_(False))
        y.sum().backward()
    def eik():
        net.zero_grad()
        net.eikonal_loss(x, 0.1).backward()
    for name, fn in (('fwd+dense bwd', main_bwd), ('eikonal', eik)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(f"S={os.environ.get('D3H_DW_S', '256'):>4s} n={n:6d} {name:14s} {(time.time() - t0) / 10 * 1e3:7.3f} ms")
