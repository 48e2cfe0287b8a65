"""whole-tick parity (oracle/parity.py) of the config-2 scene after 5 / 25 / 70 optimiser steps, with stage details -- why does the agreement
depend on how long the scene has trained?      python tools/dbg/gpu_dbg_parity_steps.py"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd')); sys.path.insert(0, ROOT)
import torch
from d3h import scene
from oracle import parity as OP
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
done = 0
for n in [int(a) for a in sys.argv[1:]] or [5, 25, 70]:
    while done < n:
        sc.step(); done += 1
    rep, tm = OP.scene_tick_parity(sc, iteration=10, seed=0, detail=True)
    det = rep.pop('detail')
    print('   grad detail', json.dumps(rep.pop('grad_detail')))
    print('==== after', n, 'steps', json.dumps({k: rep[k] for k in ('mesh_faces', 'mesh_faces_equal', 'raster_ids_differ', 'alpha_pixels_differ')}))
    for w in ('own_raster', 'shared_raster'):
        print('  ', w, json.dumps({k: rep[w][k] for k in ('max_rel_loss_diff', 'max_rel_grad_diff', 'l2_rel_grad_diff')}))
    print('   detail', json.dumps(det, indent=1)[:2500])
