"""The ORACLE's backward produced non-finite gradients on a config-2 state inside bench.py (the GPU's were finite).  Hunt: train the config-2
scene, every 10 steps snapshot the state and run the oracle tick + backward under autograd anomaly detection until it trips."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT)
import torch
from d3h import scene
from oracle import parity as OP, tick as OTK, render as ORD
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
gen = torch.Generator().manual_seed(1000)
bg = torch.rand(1, 512, 512, 3, generator=gen)
torch.manual_seed(2000)
draws = ORD.draw_jitter(1, 512, 512)
for rnd in range(int(os.environ.get('ROUNDS', 14))):
    for _ in range(10):
        sc.step()
    pts = torch.rand(50000, 3) - 0.5
    st = OP.state_from_scene(sc, bg, pts, 10)
    ro = OTK.tick_init(st, buffers=('shaded',), draws=draws, keep=True)
    for name, term in (('msk', ro['msk_loss']), ('reg', ro['reg_loss'])):
        try:
            with torch.autograd.detect_anomaly():
                term.backward(retain_graph=True)
        except RuntimeError as e:
            print('ANOMALY in', name, 'after', 10 * (rnd + 1), 'steps:', str(e)[:3000])
            torch.save({k: v for k, v in st.items() if k not in ('body',)}, os.path.join(ROOT, 'gpurun_out', 's9', 'nan_state.pt'))
            sys.exit(0)
        bad = {k: int((~torch.isfinite(v)).sum()) for k, v in OP.oracle_grads(st).items() if v is not None and not torch.isfinite(v).all()}
        print('steps', 10 * (rnd + 1), name, 'non-finite:', bad, flush=True)
