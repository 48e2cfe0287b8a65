"""Generate tests/golden/*.npz by importing the reference (dev container only) and check the oracle
against it.  Run: python tools/gen_golden.py [names...].  The fixtures are DATA (seeded inputs and the
reference's outputs); no reference source is stored.
"""
import os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, ROOT)
import refharness
from oracle import marching_tets as OMT
from oracle import sdf_mlp as OMLP

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_sdf_mlp():
    with refharness.ref_ctx():
        from geometry.mlp import MLP
        torch.manual_seed(0)
        m = MLP(skip_in=[3], n_freq=6, n_hidden=6, d_hidden=256)
        g = torch.Generator().manual_seed(1)
        x = (torch.rand(1024, 3, generator=g) * 2.4 - 1.2).requires_grad_(True)
        y = m(x)
        wgt = torch.linspace(-1, 1, 1024)[:, None]
        (y * wgt).sum().backward()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        out = {'x': x.detach(), 'sdf': y.detach(), 'gout': wgt, 'dx': x.grad}
        for k, p in m.named_parameters():
            out['grad.' + k] = p.grad
        for k, v in sd.items():
            out['sd.' + k] = v
    # oracle check
    yo = OMLP.mlp_forward(x.detach(), sd)
    assert torch.equal(yo, y.detach()) or (yo - y.detach()).abs().max() < 1e-7, (yo - y).abs().max()
    print('sdf_mlp: oracle max diff', (yo - y.detach()).abs().max().item())
    np.savez_compressed(os.path.join(GOLD, 'sdf_mlp.npz'), **npy(out))


def _mtets_inputs(n, seed, mixed_msdf, shuffle):
    verts, tets = OMT.kuhn_grid(n, shuffle_seed=(seed if shuffle else None))
    g = torch.Generator().manual_seed(seed)
    v = torch.from_numpy(verts)
    t = torch.from_numpy(tets)
    # noisy ellipsoid, positive outside
    c = torch.tensor([0.0, -0.2, 0.0])
    r = torch.tensor([0.55, 0.8, 0.45])
    sdf = (((v - c) / r).norm(dim=-1) - 1.0) * 0.4 + 0.02 * torch.randn(v.shape[0], generator=g)
    if mixed_msdf:
        msdf = torch.rand(v.shape[0], generator=g) * 2 - 0.8
    else:
        msdf = (torch.rand(v.shape[0], generator=g) - 0.01).clamp(-1, 1)      # hmsdf.py:311
    pos = v + 0.01 * torch.randn(v.shape, generator=g)
    return pos, sdf, msdf, t


def _run_ref_mtets(cls_name, pos, sdf, msdf, tets, typ=None):
    with refharness.ref_ctx():
        if cls_name == 'gshell':
            from geometry.gshell_tets import GShell_Tets as C
        else:
            from geometry.hmsdf_tets_split import hmSDF_Tets as C
        mt = C()
        pos = pos.clone().requires_grad_(True); sdf = sdf.clone().requires_grad_(True); msdf = msdf.clone().requires_grad_(True)
        if typ is None:
            verts, faces, _, _, v_tng, extra = mt(pos, sdf[:, None], msdf, tets)
        else:
            verts, faces, _, _, v_tng, extra = mt(pos, sdf[:, None], msdf, tets, typ)
        out = {'verts': verts, 'faces': faces, 'v_tng': v_tng, **{k: v for k, v in extra.items()}}
        if verts.shape[0] > 0:
            gen = torch.Generator().manual_seed(7)
            gv = torch.randn(verts.shape, generator=gen)
            gm = torch.randn(extra['msdf'].shape, generator=gen)
            gw = torch.randn(extra['vertices_watertight'].shape, generator=gen)
            loss = (verts * gv).sum() + (extra['msdf'] * gm).sum() + (extra['vertices_watertight'] * gw).sum()
            loss.backward()
            out.update({'g_verts': gv, 'g_msdf': gm, 'g_wt': gw, 'd_pos': pos.grad, 'd_sdf': sdf.grad, 'd_msdf': msdf.grad})
    return out


def gen_mtets():
    cases = [('gshell_n8', 'gshell', 8, 3, False, False, None), ('gshell_n16_mixed', 'gshell', 16, 4, True, True, None),
             ('hmsdf_n8_cloth', 'hmsdf', 8, 5, True, False, 'cloth'), ('hmsdf_n8_body', 'hmsdf', 8, 5, True, False, 'body'),
             ('gshell_n6_empty', 'gshell', 6, 6, False, False, None)]
    for name, cls, n, seed, mixed, shuffle, typ in cases:
        pos, sdf, msdf, tets = _mtets_inputs(n, seed, mixed, shuffle)
        if 'empty' in name:
            sdf = sdf.abs() + 0.1
        ref = _run_ref_mtets(cls, pos, sdf, msdf, tets, typ)
        # oracle check (bit-exact indices, tight fp)
        p2 = pos.clone().requires_grad_(True); s2 = sdf.clone().requires_grad_(True); m2 = msdf.clone().requires_grad_(True)
        o = OMT.gshell_tets(p2, s2, m2, tets, negate_msdf=(typ == 'body'))
        assert torch.equal(o['faces'], ref['faces']), name
        assert torch.equal(o['faces_watertight'], ref['faces_watertight']), name
        assert o['n_verts_watertight'] == ref['n_verts_watertight']
        for k in ('verts', 'v_tng', 'vertices_watertight', 'msdf', 'msdf_watertight', 'msdf_boundary', 'v_tng_watertight'):
            d = (o[k] - ref[k]).abs().max().item() if ref[k].numel() else 0.0
            assert d < 1e-6, (name, k, d)
        if 'd_pos' in ref:
            loss = (o['verts'] * ref['g_verts']).sum() + (o['msdf'] * ref['g_msdf']).sum() + (o['vertices_watertight'] * ref['g_wt']).sum()
            loss.backward()
            for k, t in (('d_pos', p2), ('d_sdf', s2), ('d_msdf', m2)):
                if ref[k] is None:
                    assert t.grad is None, (name, k)
                    continue
                d = (t.grad - ref[k]).abs().max().item()
                rel = d / (ref[k].abs().max().item() + 1e-12)
                assert rel < 1e-4, (name, k, d, rel)
        print(f'mtets {name}: P={ref["verts"].shape[0]} P_wt={ref["n_verts_watertight"]} F={ref["faces"].shape[0]} oracle==reference')
        out = {**{k: v for k, v in ref.items() if v is not None}, 'in_pos': pos, 'in_sdf': sdf, 'in_msdf': msdf, 'tets': tets}
        np.savez_compressed(os.path.join(GOLD, f'mtets_{name}.npz'), **npy(out))


ALL = {'sdf_mlp': gen_sdf_mlp, 'mtets': gen_mtets}

if __name__ == '__main__':
    names = sys.argv[1:] or list(ALL)
    for nme in names:
        ALL[nme]()
