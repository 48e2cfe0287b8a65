"""config-3 / config-5 / seq steps with NaN-filled torch.empty (torch.utils.deterministic.fill_uninitialized_memory): every loss must stay finite"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'd3human-code_amd'))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import warnings
warnings.filterwarnings('ignore')
from d3h.scene import Scene
for kw, fn in ((dict(res=1024, grid_n=63, n_frames=4, loss_set='full'), 'step'), (dict(res=512, grid_n=32, n_frames=2, loss_set='split'), 'step_split'),
               (dict(res=512, grid_n=32, n_frames=1, loss_set='seq'), 'step_seq')):
    sc = Scene(device='cuda', prefit_steps=300, visualize_watertight=True, **kw)
    for i in range(12):
        r = getattr(sc, fn)()
        bad = [k for k, v in r.items() if not bool(torch.isfinite(v).all())]
        assert not bad, (fn, i, bad)
    for p in list(sc.geometry.parameters()) + list(sc.material['kd_ks'].parameters()):
        assert p.grad is None or bool(torch.isfinite(p.grad).all()), fn
    print(fn, 'ok', {k: round(float(v), 5) for k, v in list(r.items())[:4]})
