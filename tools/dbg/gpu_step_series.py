"""ms per training step over the first iterations after the pre-fit (config 3): how many warm-up steps does the rate need to settle?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'd3human-code_amd'))
import torch
from d3h.scene import Scene
sc = Scene(res=1024, grid_n=63, n_frames=4, device='cuda', prefit_steps=300, loss_set='full', visualize_watertight=True)
ts = []
for k in range(24):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5):
        sc.step()
    torch.cuda.synchronize(); ts.append((time.time() - t) / 5 * 1e3)
    md = sc.geometry.last_mesh_dict
print(' '.join(f'{t:.2f}' for t in ts))
print('verts', md['imesh'].v_pos.shape[0] if 'imesh' in md else None)
