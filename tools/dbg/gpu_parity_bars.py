"""Where do the whole-tick parity figures land?  configs[1] over several seeds / states (+ optionally the config-3 shape once):
prints the shared / own figures with and without the kink exclusion.  gpurun -- 'python tools/dbg/gpu_parity_bars.py [n_seeds] [c3]'"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from d3h import scene
from oracle import parity as OP


def brief(rep):
    o = {k: rep[k] for k in ('mesh_faces', 'mesh_faces_equal', 'raster_ids_differ', 'alpha_pixels_differ', 'relu_kinks')}
    for w in ('shared_raster', 'own_raster'):
        c = rep[w]
        o[w] = {k: c.get(k) for k in ('max_rel_loss_diff', 'max_rel_grad_diff', 'l2_rel_grad_diff', 'max_rel_grad_diff_excl', 'l2_rel_grad_diff_excl',
                                      'excluded_grid_vertices', 'kink_triangles', 'alpha_pixels_differ')}
        o[w]['worst_tensors'] = sorted(((k, v) for k, v in c.get('per_tensor', {}).items()), key=lambda kv: -kv[1][0])[:4]
    return o


def kinkfree(sc):
    """a texture state without ReLU kinks: positive table entries and positive first / second layer weights keep every hidden pre-activation
    of the texture MLP (mlptexture.py:18-41: no biases) strictly positive -- the piecewise-linear network is then evaluated inside ONE linear
    piece by both implementations, whatever their summation order"""
    tex = sc.material['kd_ks']
    tex.encoder.params.data.uniform_(0.05, 0.35)
    tex.net.net[0].weight.data.abs_().add_(0.02)
    tex.net.net[2].weight.data.abs_().add_(0.02)


n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=512, grid_n=32, n_frames=1, loss_set='mask')
for s in range(n):
    for _ in range(5):
        sc.step()
    rep, _ = OP.scene_tick_parity(sc, iteration=10, seed=s)
    print('C2', s, json.dumps(brief(rep)), flush=True)
if 'c3' in sys.argv:
    del sc
    sc = scene.Scene(device='cuda', prefit_steps=300, visualize_watertight=True, res=1024, grid_n=63, n_frames=1, loss_set='full')
    for _ in range(5):
        sc.step()
    kinkfree(sc) if 'kinkfree' in sys.argv else sc.material['kd_ks'].encoder.params.data.uniform_(-0.3, 0.3)
    reps = next((int(a[5:]) for a in sys.argv if a.startswith('reps=')), 1)
    for k in range(reps):
        if k:
            for _ in range(3):
                sc.step()
            kinkfree(sc) if 'kinkfree' in sys.argv else None
        rep, _ = OP.scene_tick_parity(sc, iteration=10, seed=1 + k, detail='detail' in sys.argv)
        b = brief(rep)
        sh = b['shared_raster']
        print('C3', k, json.dumps({'faces': b['mesh_faces'], 'ids_differ': b['raster_ids_differ'], 'alpha_own': b['alpha_pixels_differ'], 'relu': b['relu_kinks'],
                                   'sh_alpha': sh['alpha_pixels_differ'], 'sh_loss': sh['max_rel_loss_diff'], 'sh_max': sh['max_rel_grad_diff_excl'],
                                   'sh_l2': sh['l2_rel_grad_diff_excl'], 'sh_excl': sh['excluded_grid_vertices'],
                                   'sh_outliers': rep['shared_raster'].get('vertex_outliers_excl'),
                                   'own_loss': b['own_raster']['max_rel_loss_diff'], 'own_max': b['own_raster']['max_rel_grad_diff_excl'],
                                   'own_l2': b['own_raster']['l2_rel_grad_diff_excl'], 'own_excl': b['own_raster']['excluded_grid_vertices']}), flush=True)
    print('C3', json.dumps(brief(rep)), flush=True)
    if 'detail' in sys.argv:
        print('C3 shared grad_detail', json.dumps(rep['shared_raster'].get('grad_detail')), flush=True)
