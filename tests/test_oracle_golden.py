"""The CPU oracle against the golden vectors produced by the reference itself (tools/gen_golden.py)."""
import glob
import os

import numpy as np
import torch

from conftest import GOLD, golden
from oracle import marching_tets as OMT
from oracle import sdf_mlp as OMLP


def test_oracle_sdf_mlp_matches_reference():
    g = golden('sdf_mlp.npz')
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = OMLP.mlp_forward(x, sd)
    assert (y.detach().numpy() - g['sdf']).__abs__().max() < 1e-7
    (y * torch.from_numpy(g['gout'])).sum().backward()
    assert np.abs(x.grad.numpy() - g['dx']).max() < 1e-5 * np.abs(g['dx']).max() + 1e-7


def test_oracle_marching_tets_matches_reference():
    for f in sorted(glob.glob(os.path.join(GOLD, 'mtets_*.npz'))):
        g = np.load(f)
        o = OMT.gshell_tets(torch.from_numpy(g['in_pos']), torch.from_numpy(g['in_sdf']), torch.from_numpy(g['in_msdf']),
                            torch.from_numpy(g['tets']), negate_msdf=('body' in f))
        assert np.array_equal(o['faces'].numpy(), g['faces']), f
        assert np.array_equal(o['faces_watertight'].numpy(), g['faces_watertight']), f
        for k in ('verts', 'v_tng', 'vertices_watertight', 'msdf', 'msdf_watertight', 'msdf_boundary'):
            if g[k].size:
                assert np.abs(o[k].detach().numpy() - g[k]).max() < 1e-6, (f, k)


def test_kuhn_grid_sizes():
    v, t = OMT.kuhn_grid(4)
    assert v.shape == (125, 3) and t.shape == (384, 4)
    # every tet has positive volume magnitude (non-degenerate)
    p = v[t]
    vol = np.abs(np.einsum('ij,ij->i', np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0]))
    assert vol.min() > 1e-6
