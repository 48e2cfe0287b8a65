"""cProfile of the host side of 200 config-3 steps (where do the ~2.3 ms of host time per step go?).  gpurun -- 'python tools/dbg/gpu_host_profile.py'"""
import cProfile
import os
import pstats
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(20):
    sc.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    sc.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(70)
st.sort_stats('tottime').print_stats(45)
