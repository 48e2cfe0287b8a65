"""Where does a non-finite gradient come from?  Config-2 scene, NaN-poisoned torch.empty (a kernel that reads a buffer it was supposed to
write completely shows up), backward of msk + reg as oracle/parity.py does, after allocator churn (other scenes built and freed)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT)
import torch
if os.environ.get('POISON', '1') == '1':
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
from d3h import scene
from oracle import parity as OP
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
for i in range(int(os.environ.get('STEPS', 20))):
    r = sc.step()
print('steps ok', {k: float(v) for k, v in r.items()})
g = sc.geometry
for trial in range(3):
    sc._zero_grad()
    r = g.tick_init(sc.glctx, sc.target(torch.rand(1, 512, 512, 3, device='cuda')), None, sc.material, sc.loss_fn, 10, None)
    tot = {0: r['msk_loss'] + r['reg_loss'], 1: r['msk_loss'], 2: r['reg_loss']}[trial]
    tot.backward()
    bad = {k: int((~torch.isfinite(v)).sum()) for k, v in OP.scene_grads(sc).items() if v is not None and not torch.isfinite(v).all()}
    print('trial', trial, 'total', float(tot), 'non-finite grads:', bad)
