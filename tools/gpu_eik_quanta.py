"""upper bound of what removing the 3.05 -> 4 tile-quantum loss of the 50 000-sample eikonal sweeps can give: the config-3 step with 49 152
samples (exactly three 16-point tiles per SIMD) against 50 000 and 52 000.   python tools/gpu_eik_quanta.py"""
import os, sys, time, gc
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'd3human-code_amd'))
import torch
from d3h import scene
for ns in (50000, 49152, 50000, 49152, 52000):
    sc = scene.Scene(device='cuda:0', prefit_steps=300, visualize_watertight=True, res=1024, grid_n=63, n_frames=4, loss_set='full',
                     flags_hook=lambda F: setattr(F, 'eikonal_samples', ns))
    for _ in range(15):
        sc.step()
    torch.cuda.synchronize(); gc.collect(); gc.freeze()
    t0 = time.time()
    for _ in range(100):
        sc.step()
    torch.cuda.synchronize()
    print(f'eikonal samples {ns}: {(time.time() - t0) * 10:.3f} ms/step', flush=True)
    del sc
