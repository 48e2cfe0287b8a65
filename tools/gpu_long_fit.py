"""A long config-3 fit on the GPU: losses, mesh size and rate every 250 iterations (is the workload stable, does the fit converge?).
gpurun -- 'python tools/gpu_long_fit.py > gpurun_out/long_fit.txt'"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene
N = int(os.environ.get('ITERS', 3000))
sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
print('iter   it/s   msk_loss  img_loss  normal    sdf_reg   eik       max|g|(sdf net)  mesh verts / faces')
t = time.time()
for it in range(1, N + 1):
    r = sc.step()
    if it % 250 == 0 or it in (1, 50):
        gmax = max(float(p.grad.abs().max()) for p in sc.geometry.sdf_net.parameters() if p.grad is not None)
        torch.cuda.synchronize()
        dt = time.time() - t
        md = sc.geometry.last_mesh_dict
        v, f = md['imesh'].v_pos.shape[0], md['imesh'].t_pos_idx.shape[0]
        g = lambda k: float(r[k]) if k in r else float('nan')
        print(f'{it:5d} {250 / dt if it % 250 == 0 else float("nan"):6.1f}  {g("msk_loss"):8.4f}  {g("img_loss"):8.4f}  {g("normal_loss"):8.4f}  '
              f'{g("sdf_reg_loss"):8.4f}  {g("eik_loss"):8.5f}  {gmax:10.3e}      {v} / {f}', flush=True)
        t = time.time()
