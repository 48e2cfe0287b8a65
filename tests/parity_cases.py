"""Parity checks shared by the emulated (CPU, toy sizes) and the real (-m gpu) runs of the kernels.

Every check drives the product's Python wrappers (d3h.*), which call the C ABI; `dev` is 'cuda' for the
HIP library and 'cpu' only under the test-only emulator hook.
"""
import glob
import os

import numpy as np
import torch

from conftest import GOLD, golden


def T(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def sd_from_golden(g, dev):
    return {k[3:]: T(g[k], dev) for k in g.files if k.startswith('sd.')}


# ---- SDF MLP -----------------------------------------------------------------------------------
def check_sdf_mlp_forward(dev, n=None, tol=2e-7, x3=False):
    """HIP fused PE+MLP vs the reference MLP's own outputs (golden) -- fp32 tolerance, sign agreement.  x3: the DEFAULT arithmetic of the
    product (bf16 x 3 operand split on the bf16 matrix pipe, csrc/sdf_mlp_x3.h) instead of the exact-f32 MFMA kernel."""
    from d3h import sdf_mlp
    g = golden('sdf_mlp.npz')
    sd = sd_from_golden(g, dev)
    x = T(g['x'], dev)
    ref = g['sdf'].reshape(-1)
    if n is not None:
        x, ref = x[:n].contiguous(), ref[:n]
    wp = sdf_mlp.pack_weights(sd)
    out = sdf_mlp.forward(x, wp, wp3=sdf_mlp.pack_weights3(sd) if x3 else None).cpu().numpy()
    err = np.abs(out - ref).max()
    assert err < tol, f'sdf max abs err {err}'
    # topology is decided by sdf > 0 (gshell_tets.py:260): signs must agree wherever |sdf_ref| > tau
    tau = 1e-6
    m = np.abs(ref) > tau
    assert np.array_equal(out[m] > 0, ref[m] > 0)
    return err


# ---- marching tets ---------------------------------------------------------------------------------
def check_gshell_tangents_golden(dev):
    """GShell_Tets(compute_tangents=True): v_tng / v_tng_watertight against the reference's outputs (incl. its vertex-id uv lookup quirk)"""
    from geometry.gshell_tets import GShell_Tets
    for name in ('mtets_gshell_n8.npz', 'mtets_gshell_n16_mixed.npz'):
        g = golden(name)
        mt = GShell_Tets()
        mt.compute_tangents = True
        verts, faces, _, _, v_tng, extra = mt(T(g['in_pos'], dev), T(g['in_sdf'], dev), T(g['in_msdf'], dev), T(g['tets'], dev))
        assert np.array_equal(faces.cpu().numpy(), g['faces'])
        assert np.abs(extra['v_tng_watertight'].cpu().numpy() - g['v_tng_watertight']).max() < 2e-5, name
        assert np.abs(v_tng.cpu().numpy() - g['v_tng']).max() < 2e-5, name


def check_mtets_speculative(dev, n=9):
    """d3h.mtets: the speculative extraction (emit kernels queued at the previous sizes x 1.25 + 256 before the host reads the sizes) returns
    exactly what the exact-size path returns -- values, indices, shapes, gradients -- when the capacities hold AND when the surface outgrows
    them (nothing written, repeated at the exact sizes), for the garment and the body pass of one grid, through several changes of the field."""
    from d3h import mtets, synth
    verts, tets = (torch.from_numpy(a) for a in synth.kuhn_grid(n))
    verts, tets = verts.to(dev), tets.to(dev)
    gen = torch.Generator().manual_seed(3)

    def field(r, wob):
        c = torch.tensor([0.02, -0.03, 0.01], device=dev)
        return ((verts - c).norm(dim=-1) - r + wob * torch.sin(7.0 * verts[:, 0]) * torch.cos(5.0 * verts[:, 1])).contiguous()

    def run(sdf, msdf, body, speculate):
        mtets.SPECULATE = speculate
        pos = verts.clone().requires_grad_(True)
        s_ = sdf.clone().requires_grad_(True)
        m_ = msdf.clone().requires_grad_(True)
        o = mtets.marching_tets(pos, s_, m_, tets, body=body)
        w = torch.arange(o['verts'].numel(), device=dev, dtype=torch.float32).reshape(o['verts'].shape).sin()
        ((o['verts'] * w).sum() + (o['msdf'] * 0.3).sum() + (o['verts_wt'] * 0.7).sum()).backward()
        return o, (pos.grad, s_.grad, m_.grad)

    keep = mtets.SPECULATE
    try:
        mtets.TetGrid._cache.clear()
        stats0 = dict(mtets.SPEC_STATS)
        cases = [(0.30, 0.00), (0.31, 0.01), (0.33, 0.02), (0.80, 0.05), (0.34, 0.02), (0.12, 0.0), (0.85, 0.08)]     # grows past 1.25x + 256 twice
        for it, (r, wob) in enumerate(cases):
            sdf = field(r, wob)
            msdf = (torch.rand(verts.shape[0], generator=gen).to(dev) - 0.35).contiguous()
            for body in (False, True):
                o_s, g_s = run(sdf, msdf, body, True)
                grid = mtets.TetGrid.get(tets)
                caps = dict(grid.caps)
                o_e, g_e = run(sdf, msdf, body, False)
                grid.caps = caps                      # (the exact run must not move the capacities the next speculative run starts from)
                for k in ('verts', 'faces', 'verts_wt', 'faces_wt', 'msdf', 'faces32', 'faces_wt32', 'bnd_edge'):
                    assert o_s[k].shape == o_e[k].shape and torch.equal(o_s[k], o_e[k]), (it, body, k)
                    assert o_s[k].is_contiguous()
                assert o_s['n_wt'] == o_e['n_wt']
                for a, b in zip(g_s, g_e):
                    assert (a is None) == (b is None)
                    if a is not None:                 # float atomics: same addends, another order
                        assert (a - b).abs().max() <= 1e-5 * max(1e-12, float(b.abs().max())), (it, body)
        d = {k: mtets.SPEC_STATS[k] - stats0[k] for k in stats0}
        assert d['speculated'] == 2 * (len(cases) - 1) and d['overflowed'] >= 2, d
    finally:
        mtets.SPECULATE = keep


def check_mtets_golden(dev, names=None):
    """bit-exact indices + fp values, gradients to 1e-5 relative, against the reference's own outputs."""
    from d3h import mtets
    files = sorted(glob.glob(os.path.join(GOLD, 'mtets_*.npz')))
    for f in files:
        name = os.path.basename(f)
        if names and not any(n in name for n in names):
            continue
        g = np.load(f)
        pos, sdf, msdf = T(g['in_pos'], dev, True), T(g['in_sdf'], dev, True), T(g['in_msdf'], dev, True)
        tets = T(g['tets'], dev)
        seen = []
        o = mtets.marching_tets(pos, sdf, msdf, tets, body=('body' in name), before_face_sync=lambda v, vw, fp: seen.append((v.shape, vw.shape, fp.shape[0] >= 0)))
        # the vertex-only hook runs once, with the final vertex tensors, before the face list is narrowed from its 2 n1 + 4 n2 bound
        assert seen == [(o['verts'].shape, o['verts_wt'].shape, True)], name
        assert o['faces'].dtype == torch.int64
        assert np.array_equal(o['faces'].cpu().numpy(), g['faces']), name
        assert np.array_equal(o['faces_wt'].cpu().numpy(), g['faces_watertight']), name
        assert o['n_wt'] == int(g['n_verts_watertight']), name
        assert np.array_equal(o['faces32'].cpu().numpy().astype(np.int64), g['faces']), name
        for k, gk in (('verts', 'verts'), ('verts_wt', 'vertices_watertight'), ('msdf', 'msdf')):
            a = o[k].detach().cpu().numpy()
            assert a.shape == g[gk].shape, (name, k)
            if a.size:
                assert np.abs(a - g[gk]).max() <= 1e-7, (name, k, np.abs(a - g[gk]).max())
        if 'd_pos' in g.files:
            loss = (o['verts'] * T(g['g_verts'], dev)).sum() + (o['msdf'] * T(g['g_msdf'], dev)).sum() + \
                (o['verts_wt'] * T(g['g_wt'], dev)).sum()
            loss.backward()
            for k, t in (('d_pos', pos), ('d_sdf', sdf), ('d_msdf', msdf)):
                if k in g.files:
                    d = np.abs(t.grad.cpu().numpy() - g[k]).max() / (np.abs(g[k]).max() + 1e-12)
                    assert d < 1e-5, (name, k, d)
                else:
                    assert t.grad is None, (name, k)      # "body": msdf negated under no_grad in the reference


def check_sdf_mlp_backward(dev, n=None, tol=2e-5, sparse_gout=False):
    """fused backward (dx, all 16 parameter grads) vs the reference's autograd (golden) or the oracle's on a subset.
    sparse_gout: zero the upstream gradient on most 16-point tiles (what a training sweep produces: the loss reads the sdf only next
    to the surface) -- exercises the active-tile list of the backward, including an odd tile count and a ragged last tile"""
    from d3h import sdf_mlp
    from oracle import sdf_mlp as O
    g = golden('sdf_mlp.npz')
    keys = sdf_mlp._PARAM_ORDER
    params = [T(g['sd.net.' + k], dev, True) for k in keys]
    xs = g['x'] if n is None else g['x'][:n]
    go = g['gout'] if n is None else g['gout'][:n]
    if sparse_gout:
        assert n is not None
        go = go.copy()
        keep = np.zeros((n + 15) // 16, bool)
        keep[[1, 4, (n + 15) // 16 - 1]] = True                  # three active tiles (odd), the last one ragged when n % 16 != 0
        m = np.repeat(keep, 16)[:n]
        go[~m] = 0.0
        go[m] = np.where(np.arange(m.sum())[:, None] % 3 == 0, 0.0, go[m] + 0.37)    # zeros inside active tiles too
    x = T(xs, dev, True)
    y = sdf_mlp.sdf_query(x, params)
    (y * T(go, dev)).sum().backward()
    if n is None:
        ref_dx = g['dx']
        ref = {k: g['grad.net.' + k] for k in keys}
    else:
        sd = {('net.' + k): torch.from_numpy(g['sd.net.' + k]).requires_grad_(True) for k in keys}
        x2 = torch.from_numpy(xs).requires_grad_(True)
        (O.mlp_forward(x2, sd) * torch.from_numpy(go)).sum().backward()
        ref_dx = x2.grad.numpy()
        ref = {k: sd['net.' + k].grad.numpy() for k in keys}
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-20)
    assert rel(x.grad.cpu().numpy(), ref_dx) < tol
    for k, p in zip(keys, params):
        if k == '14.bias':      # = sum(gout), ~0 for the golden's antisymmetric weights: absolute tolerance
            assert abs(p.grad.item() - float(ref[k].reshape(-1)[0])) < 1e-5 * np.abs(go).sum(), k
        else:
            assert rel(p.grad.cpu().numpy(), ref[k]) < tol, k


def check_sdf_mlp_eikonal(dev, n=200, tol=5e-5, scale=1.0):
    """hand-derived second-order pass (d3h.sdf_mlp.sdf_gradient) vs torch double backward through the oracle MLP
    (the reference's eikonal term: hmsdf.py:856-876)"""
    from d3h import sdf_mlp
    from oracle import sdf_mlp as O
    g = golden('sdf_mlp.npz')
    keys = sdf_mlp._PARAM_ORDER
    # the golden's weights give |grad f| ~ 1e-2..1e-1; `scale` on the head keeps the loss gradient well-conditioned
    vals = {k: g['sd.net.' + k].copy() for k in keys}
    vals['14.weight'] *= scale
    params = [T(vals[k], dev, True) for k in keys]
    xs = np.ascontiguousarray(g['x'][:n])
    gr = sdf_mlp.sdf_gradient(T(xs, dev), params)
    loss = 0.3 * (gr.pow(2).sum(dim=-1).sqrt() - 1).pow(2).mean()
    loss.backward()
    sd = {('net.' + k): torch.from_numpy(vals[k]).requires_grad_(True) for k in keys}
    x2 = torch.from_numpy(xs).requires_grad_(True)
    g2 = torch.autograd.grad(O.mlp_forward(x2, sd).sum(), x2, create_graph=True)[0]
    loss2 = 0.3 * (g2.pow(2).sum(dim=-1).sqrt() - 1).pow(2).mean()
    loss2.backward()
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-20)
    assert rel(gr.detach().cpu().numpy(), g2.detach().numpy()) < 2e-5
    assert abs(loss.item() - loss2.item()) < 1e-5 * abs(loss2.item())
    worst = 0.0
    for k, p in zip(keys, params):
        if k == '14.bias':
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
            continue
        r = rel(p.grad.cpu().numpy(), sd['net.' + k].grad.numpy())
        worst = max(worst, r)
        assert r < tol, (k, r)
    # the one-op form with eagerly computed gradients (what HmSDFTetsGeometry._eikonal calls), scaled by an upstream factor
    params2 = [T(vals[k], dev, True) for k in keys]
    l3 = sdf_mlp.eikonal_loss(T(xs, dev), params2, 0.3)
    assert abs(l3.item() - loss2.item()) < 1e-5 * abs(loss2.item())
    (l3 * 1.7).backward()
    for k, p in zip(keys, params2):
        if k == '14.bias':
            continue
        r = rel(p.grad.cpu().numpy() / 1.7, sd['net.' + k].grad.numpy())
        assert r < tol, ('eager', k, r)
    return worst


def check_sdf_mlp_deform(dev, n=300):
    """x = verts + disp*deform path (hmsdf.py:433): forward value, xdef, and d(deform) = disp * dx"""
    from d3h import sdf_mlp
    from oracle import sdf_mlp as O
    g = golden('sdf_mlp.npz')
    keys = sdf_mlp._PARAM_ORDER
    params = [T(g['sd.net.' + k], dev) for k in keys]
    gen = torch.Generator().manual_seed(3)
    verts = torch.from_numpy(g['x'][:n])
    deform = (torch.rand(n, 3, generator=gen) * 2 - 1)
    disp = 1.0 / 128 / 2.1
    d = deform.clone().to(dev).requires_grad_(True)
    y = sdf_mlp.sdf_query(verts.to(dev), params, deform=d, disp=disp)
    y.sum().backward()
    sd = {('net.' + k): torch.from_numpy(g['sd.net.' + k]) for k in keys}
    d2 = deform.clone().requires_grad_(True)
    v2, y2 = O.sdf_sweep(verts, d2, disp, sd)
    y2.sum().backward()
    assert (y.detach().cpu() - y2.detach()).abs().max() < 2e-7
    assert (d.grad.cpu() - d2.grad).abs().max() / d2.grad.abs().max() < 2e-5


# ---- seq-stage geometry terms ---------------------------------------------------------------------------
def check_seq_ops_golden(dev):
    """collision / uniform-Laplacian / normal-consistency losses + connected faces vs the reference's own outputs (seq.npz)"""
    from d3h import meshops as M
    g = golden('seq.npz')
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    # topology helpers: bit-exact
    all_f = T(g['all_f'], dev)
    pairs, _ = M.find_connected_faces(all_f)
    assert np.array_equal(pairs.cpu().numpy(), g['connected_faces'])
    assert np.array_equal(M.find_edges(all_f).cpu().numpy(), g['edges_unique'])
    # collision
    c, b = T(g['cloth_v'], dev, True), T(g['body_v'], dev, True)
    bf = T(g['body_f'], dev)
    l = M.collision_loss(c, b, bf, push_eps=float(g['colli_eps']))
    assert abs(l.item() - float(g['colli'])) < 1e-6 * abs(float(g['colli']))
    (l * 3.0).backward()
    assert rel(c.grad.cpu().numpy() / 3.0, g['colli_dcloth']) < 2e-5
    assert rel(b.grad.cpu().numpy() / 3.0, g['colli_dbody']) < 2e-5
    l0 = M.collision_loss(T(g['cloth_v'], dev), T(g['body_v'], dev), bf)
    assert abs(l0.item() - float(g['colli_default'])) <= 1e-6 * abs(float(g['colli_default'])) + 1e-12
    # uniform Laplacian
    v = T(g['all_v'], dev, True)
    ll = M.laplacian_loss(v, T(g['mesh_edges'], dev))
    assert abs(ll.item() - float(g['lap'])) < 2e-6 * float(g['lap'])
    ll.backward()
    assert rel(v.grad.cpu().numpy(), g['lap_dv']) < 2e-5
    # normal consistency
    v2 = T(g['all_v'], dev, True)
    nl = M.normal_consistency(v2, all_f.int().contiguous(), pairs.int().contiguous())
    assert abs(nl.item() - float(g['ncons'])) < 1e-5 * float(g['ncons'])
    nl.backward()
    assert rel(v2.grad.cpu().numpy(), g['ncons_dv']) < 5e-5


def check_mesh_sdf(dev, n=600):
    """GPU point-to-mesh signed distance (pre-fit target, replaces pysdf) vs the float64 brute-force oracle on the golden's closed
    body mesh, plus the analytic check on a sphere: |sdf - (|x| - 1)| bounded by the faceting error, signs exact away from the surface"""
    from d3h import meshops as M
    from oracle import seq_ops as O
    g = golden('seq.npz')
    v, f = torch.from_numpy(g['body_v']), torch.from_numpy(g['body_f'])
    gen = torch.Generator().manual_seed(4)
    pts = torch.cat([(torch.rand(n, 3, generator=gen) * 2 - 1) * torch.tensor([0.9, 1.2, 0.8]), v[:40] * 1.0, v[:40] * 0.5, v[40:] * 1.3])
    ref = O.mesh_sdf(pts, v, f)
    out = M.mesh_sdf(pts.to(dev), v.to(dev), f.to(dev)).cpu()
    assert (out.abs() - ref.abs()).abs().max() < 2e-6
    far = ref.abs() > 1e-4
    assert torch.equal(torch.sign(out[far]), torch.sign(ref[far]))
    assert (ref < 0).sum() > 20 and (ref > 0).sum() > 20
    # unit sphere (subdivided octahedron, 66 vertices): the mesh is inscribed, so -max sagitta (~0.07) <= (|x| - 1) - sdf <= 0
    sv = torch.nn.functional.normalize(torch.from_numpy(g['body_v']) / torch.tensor([0.5, 0.8, 0.4]), dim=-1)
    q = torch.randn(400, 3, generator=gen) * 0.8
    d = M.mesh_sdf(q.to(dev), sv.to(dev), f.to(dev)).cpu()
    err = (q.norm(dim=-1) - 1.0) - d
    assert err.min() > -0.09 and err.max() < 0.02, (float(err.min()), float(err.max()))


def check_mesh_api_seq(dev):
    """render.mesh.Mesh.laplacian / normal_consistency() and lap_loss.* through the reference's API names"""
    import lap_loss
    from render import mesh as rmesh
    g = golden('seq.npz')
    all_f = T(g['all_f'], dev)
    conn, _ = rmesh.find_connected_faces(all_f)
    v = T(g['all_v'], dev, True)
    m = rmesh.Mesh(v, all_f, connected_faces=conn)
    assert np.array_equal(m.edges.cpu().numpy(), g['mesh_edges'])
    l1 = lap_loss.body_laplacian_loss(m)
    l2 = lap_loss.body_normal_loss(m)
    assert abs(l1.item() - float(g['lap'])) < 2e-6 * float(g['lap'])
    assert abs(l2.item() - float(g['ncons'])) < 1e-5 * float(g['ncons'])
    L = m.laplacian                                   # sparse [V,V], as compute_laplacian_uniform
    lv = torch.sparse.mm(L, v.detach()) if L.is_sparse else L @ v.detach()
    assert abs(lv.norm(dim=1).pow(2).mean().item() - float(g['lap'])) < 1e-5 * float(g['lap'])


def check_mlp_deform_golden(dev):
    from geometry.mlp import MLP_deform
    g = golden('seq.npz')
    net = MLP_deform(skip_in=[3], n_freq=8, n_hidden=6, d_hidden=256, d_out=3)
    net.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('nr_sd.')})
    net = net.to(dev)
    code = T(g['nr_code'], dev, True)
    y = net(T(g['nr_x'], dev), code)
    assert (y.detach().cpu() - torch.from_numpy(g['nr_y'])).abs().max() < 2e-6
    (y * T(g['nr_w'], dev)).sum().backward()
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(code.grad.cpu().numpy(), g['nr_dcode']) < 2e-5
    for k, p in net.named_parameters():
        assert rel(p.grad.cpu().numpy(), g['nr_grad.' + k]) < 5e-5, k


def check_mlp_deform_fused_vs_library(dev, n=3000):
    """fused offset-network kernels (csrc/deform_mlp*.hip) against the library-GEMM formulation of the same module on a ragged point
    count: outputs, d(code), every parameter gradient"""
    from geometry.mlp import MLP_deform
    torch.manual_seed(3)
    net = MLP_deform(skip_in=[3], n_freq=8, n_hidden=6, d_hidden=256, d_out=3).to(dev)
    assert net.fused
    x = (torch.rand(1, n, 3, device=dev) * 2 - 1) * 0.9
    w = torch.randn(1, n, 3, device=dev)
    code_a = (torch.randn(1, 1, 136, device=dev) * 0.3).requires_grad_(True)
    code_b = code_a.detach().clone().requires_grad_(True)
    ya = net(x, code_a)
    (ya * w).sum().backward()
    ga = {k: p.grad.clone() for k, p in net.named_parameters()}
    net.zero_grad()
    yb = net.forward_reference(x, code_b)
    (yb * w).sum().backward()
    assert (ya - yb).abs().max() < 5e-6 * max(1.0, float(yb.detach().abs().max()))
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(code_a.grad, code_b.grad) < 1e-4
    for k, p in net.named_parameters():
        assert rel(ga[k], p.grad) < 1e-4, (k, rel(ga[k], p.grad))


# ---- LBS -----------------------------------------------------------------------------------------------
def _lbs_setup(dev):
    from deform.smplx_exavatar_deformer import SMPLX_Deformer
    g = golden('lbs.npz')
    md = {k[6:]: g[k] for k in g.files if k.startswith('model.')}
    md['posedirs'] = np.zeros((54 * 9, md['v_template'].shape[0] * 3), np.float32)   # does not reach A
    d = SMPLX_Deformer(model_dict=md, device=dev, shape_param_dim=10, expr_param_dim=5)
    return g, d


def check_knn_grid(dev, nv=1500, nq=3000, seed=0):
    """d3h_knn1_grid == d3h_knn1 bit for bit (indices and squared distances): surface-like template with duplicated vertices (ties
    resolve to the lowest index), queries near the template, far outside its box, exactly on vertices, non-finite; plus the
    exhaustive oracle (first minimum of the sequential scan) on a subset"""
    from d3h import lbs as HL
    import oracle.lbs as OL
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(nv, 3, generator=g)
    tmpl = u / u.norm(dim=1, keepdim=True) * torch.tensor([0.3, 0.8, 0.2]) + torch.tensor([0.0, -0.3, 0.05])
    tmpl[nv // 2:nv // 2 + 40] = tmpl[:40]                                    # exact duplicates: the lower index must win
    tmpl = tmpl.to(dev).contiguous()
    near = tmpl[torch.randint(0, nv, (nq,), generator=g).to(dev)] + 0.03 * torch.randn(nq, 3, generator=g).to(dev)
    far = (torch.rand(200, 3, generator=g).to(dev) * 2 - 1) * 5.0
    onv = tmpl[:100].clone()
    bad = torch.tensor([[float('nan'), 0, 0], [float('inf'), 0, 0], [0, float('-inf'), 0]], device=dev)
    mid = 0.5 * (tmpl[:50] + tmpl[50:100])
    pts = torch.cat([near, far, onv, bad, mid]).contiguous()
    grid = HL.KnnGrid(tmpl)
    i_g, d_g = grid.query(pts, want_dist=True)
    i_e = HL.knn1(pts, tmpl)
    assert torch.equal(i_g.cpu(), i_e.cpu())
    sub = torch.arange(0, pts.shape[0], max(1, pts.shape[0] * nv // 4_000_000))          # oracle: bounded [q, nv] distance matrix
    sub = torch.cat([sub, torch.arange(nq, pts.shape[0])]).unique()
    ref_i, ref_d = OL.knn1(pts.cpu()[sub], tmpl.cpu(), return_dist=True)
    assert torch.equal(i_g.cpu().long()[sub], ref_i.long())
    fin = torch.isfinite(ref_d)
    assert torch.equal(d_g.cpu()[sub][fin], ref_d[fin])
    assert int(i_g[nq + 200:nq + 240].max()) < nv // 2                        # duplicates -> first copy
    for tiny in (1, 2, 17):                                                   # degenerate templates (single cell / flat boxes)
        t2 = tmpl[:tiny].contiguous()
        assert torch.equal(HL.KnnGrid(t2).query(pts).cpu(), HL.knn1(pts, t2).cpu())
    flat = tmpl.clone(); flat[:, 2] = 0.25
    assert torch.equal(HL.KnnGrid(flat).query(pts).cpu(), HL.knn1(pts, flat).cpu())
    assert HL.KnnGrid(tmpl).query(pts[:0]).numel() == 0


def check_lbs_golden(dev):
    """nearest ids exact; A0/A, canonical and posed points + grads (pts, trans, body/root pose) vs the reference functions"""
    from d3h import lbs as HL
    g, d = _lbs_setup(dev)
    # initialize() would need posedirs for vs_template; the golden carries the reference's template/A0
    d.vs_template = T(g['tmpl'], dev)[None]
    betas = T(g['betas'], dev)
    z = lambda n: torch.zeros(1, n, device=dev)
    body0 = z(63); body0[:, 2] = torch.pi / 36; body0[:, 5] = -torch.pi / 36
    A0 = d.layer.transforms(betas, z(3), body0, z(3), z(5))
    assert (A0[0].cpu() - torch.from_numpy(g['A0'])).abs().max() < 2e-6
    d.init_A = A0
    pts = T(g['pts'], dev, True)
    idx = HL.knn1(pts, d.vs_template[0])
    # the reference's w_pts rows must equal the gathered rows of our nearest ids
    assert np.array_equal(d.lbs_weights[idx.long()].cpu().numpy(), g['w_pts'])
    nfr = g['out'].shape[0]
    param = {'shape': betas, 'face_offset': T(g['face_offset'], dev), 'joint_offset': T(g['joint_offset'], dev),
             'locator_offset': T(g['locator_offset'], dev), 'trans': T(g['trans'], dev, True), 'jaw_pose': T(g['jaw'], dev),
             'expr': T(g['expr'], dev), 'body_pose': T(g['body_pose'], dev, True), 'root_pose': T(g['root_pose'], dev, True)}
    A, _ = d.frame_transforms(param, range(nfr))
    assert (A.detach().cpu() - torch.from_numpy(g['A'])).abs().max() < 2e-6
    out = d.lbs_forward_batch(pts, param, range(nfr))
    assert (out.detach().cpu() - torch.from_numpy(g['out'])).abs().max() < 5e-6
    one = d.lbs_forward(pts.detach().reshape(1, -1, 3), param, idx=1)
    assert (one.detach().cpu() - torch.from_numpy(g['out'][1])).abs().max() < 5e-6
    (out * T(g['gout'], dev)).sum().backward()
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert rel(pts.grad.cpu().numpy(), g['d_pts']) < 1e-4
    assert rel(param['trans'].grad.cpu().numpy(), g['d_trans']) < 1e-4
    assert rel(param['body_pose'].grad.cpu().numpy(), g['d_body_pose']) < 1e-4
    assert rel(param['root_pose'].grad.cpu().numpy(), g['d_root_pose']) < 1e-4


# ---- rasterize / interpolate / antialias / texture ---------------------------------------------------------
def _raster_scene(res=48, nb=2, big=False):
    """marching-tets golden mesh (open boundaries + watertight interior) seen by the synthetic camera, nb slightly
    different rigid placements; or two large intersecting triangles (workgroup path of the rasterizer)"""
    from d3h import synth
    if big:
        v = np.array([[-0.9, -0.8, 0.2], [0.9, -0.7, -0.3], [0.0, 0.9, 0.1], [-0.8, 0.7, -0.2], [0.8, 0.8, 0.3], [0.1, -0.9, 0.0]], np.float32)
        f = np.array([[0, 1, 2], [3, 5, 4]], np.int64)
    else:
        g = golden('mtets_gshell_n8.npz')
        v, f = g['verts'].astype(np.float32), g['faces']
    mv, mvp, campos = synth.camera(res, dist=3.0)
    vs = []
    for b in range(nb):
        ang = 0.3 * b
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        vw = v @ R.T * (1.2 if not big else 1.0) + np.array([0.02 * b, 0.0, 0.0], np.float32)
        vh = np.concatenate([vw, np.ones((len(vw), 1), np.float32)], 1)
        vs.append(vh @ mvp.T)
    return np.stack(vs).astype(np.float32), f


def check_rasterize(dev, res=48, big=False, nb=2):
    from d3h import raster
    from oracle import raster as OR
    posn, f = _raster_scene(res, nb, big)
    pos = T(posn, dev, True)
    tri = T(f.astype(np.int32), dev)
    rast, db = raster.rasterize(pos, tri, (res, res))
    pos_o = torch.from_numpy(posn).requires_grad_(True)
    rast_o, db_o = OR.rasterize(pos_o, torch.from_numpy(f), res, res)
    r, ro = rast.detach().cpu(), rast_o.detach()
    assert torch.equal(r[..., 3], ro[..., 3]), f'{(r[..., 3] != ro[..., 3]).sum().item()} pixels differ in triangle id'
    assert (r[..., 3] > 0).float().mean() > 0.05
    assert (r[..., :3] - ro[..., :3]).abs().max() < 2e-4
    assert (db.cpu() - db_o).abs().max() < 5e-3 * db_o.abs().max()
    gen = torch.Generator().manual_seed(5)
    G = torch.randn(r.shape, generator=gen)
    (rast * G.to(dev)).sum().backward()
    (rast_o * G).sum().backward()
    rel = (pos.grad.cpu() - pos_o.grad).abs().max() / pos_o.grad.abs().max()
    assert rel < 2e-3, rel



def check_rasterize_binned(dev, res=80, n_small=3000):
    """the tile-binned rasteriser (csrc/raster.hip: raster_tile_kernel; north_star's "tile-binned differentiable rasterizer") against the
    wave-per-triangle kernels on the same inputs: bit-identical rast / rast_db, for (i) the marching-tets golden mesh in two placements, (ii) the
    triangles crossing the camera plane, (iii) a soup of small triangles (the regime the binned path is for) with two tile-sized ones on top,
    at a resolution that is a multiple of no tile (80 = 2.5 tiles of 32), and (iv) a scratch too small for the pairs: the device falls back by
    itself.  The existing parity checks of the rasteriser also run through the binned path (against oracle/raster.py)."""
    from d3h import raster

    def both(pos, tri, r):
        old = raster.BIN_MIN_TRIS
        try:
            raster.BIN_MIN_TRIS = 1 << 30
            a = raster.rasterize(pos, tri, (r, r))
            raster.BIN_MIN_TRIS = 1
            b = raster.rasterize(pos, tri, (r, r))
        finally:
            raster.BIN_MIN_TRIS = old
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), int((a[0] != b[0]).sum())
        return a[0]

    posn, f = _raster_scene(res, 2)
    r = both(T(posn, dev), T(f.astype(np.int32), dev), res)
    assert (r[..., 3] > 0).float().mean() > 0.05
    # (ii) camera-plane crossing triangles of check_rasterize_near_plane
    n, fa = 0.1, 10.0
    P = np.array([[1.2, 0, 0, 0], [0, 1.2, 0, 0], [0, 0, -(fa + n) / (fa - n), -2 * fa * n / (fa - n)], [0, 0, -1, 0]], np.float32)
    v = np.array([[-0.5, -0.4, -2.0], [0.5, -0.4, -2.0], [0.6, -0.2, 1.0], [-0.6, -0.2, 1.0], [0, 0.3, -1.5], [0.3, 0.5, -1.5], [-0.3, 0.5, -1.5],
                  [0.2, 0.1, -3.0], [0.9, 0.1, 0.5], [0.9, 0.6, -3.0]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9]], np.int32)
    vh = (np.concatenate([v, np.ones((len(v), 1), np.float32)], 1) @ P.T)[None].astype(np.float32)
    r = both(T(vh, dev), T(tri, dev), res)
    assert all(int((r[..., 3] == t).sum()) > 10 for t in (1, 2, 3, 4))
    # (iii) small-triangle soup + two large triangles, clip space with w = 1 and a spread of depths
    rng = np.random.default_rng(3)
    c = rng.uniform(-1.05, 1.05, (n_small, 1, 2)).astype(np.float32)
    tv = np.concatenate([c + rng.uniform(-0.04, 0.04, (n_small, 3, 2)).astype(np.float32), rng.uniform(-0.9, 0.9, (n_small, 3, 1)).astype(np.float32),
                         np.ones((n_small, 3, 1), np.float32)], -1).reshape(-1, 4)
    bigv = np.array([[-0.9, -0.8, 0.2, 1], [0.9, -0.7, -0.3, 1], [0.0, 0.9, 0.1, 1], [-0.8, 0.7, -0.2, 1], [0.8, 0.8, 0.3, 1], [0.1, -0.9, 0.0, 1]], np.float32)
    pos = np.concatenate([tv, bigv])[None]
    tri = np.concatenate([np.arange(3 * n_small, dtype=np.int32).reshape(-1, 3), 3 * n_small + np.array([[0, 1, 2], [3, 5, 4]], np.int32)])
    r = both(T(pos, dev), T(tri, dev), res)
    assert (r[..., 3] > 0).float().mean() > 0.5 and len(torch.unique(r[..., 3])) > 200
    # (iv) not enough room for the pairs: the flag kernel hands the render back to the wave-per-triangle path on the device
    old = raster.BIN_MIN_TRIS, raster.BIN_PAIRS_PER_TRI
    try:
        raster.BIN_MIN_TRIS, raster.BIN_PAIRS_PER_TRI = 1, 0
        b = raster.rasterize(T(pos, dev), T(tri, dev), (res, res))
    finally:
        raster.BIN_MIN_TRIS, raster.BIN_PAIRS_PER_TRI = old
    assert torch.equal(b[0], r)


def check_interpolate(dev, res=40):
    from d3h import raster
    from oracle import raster as OR
    posn, f = _raster_scene(res, 2)
    V = posn.shape[1]
    gen = torch.Generator().manual_seed(6)
    tri_o = torch.from_numpy(f)
    rast_o, db_o = OR.rasterize(torch.from_numpy(posn), tri_o, res, res)
    for batched in (False, True):
        attrn = torch.randn(2 if batched else 1, V, 5, generator=gen)
        attr, attr_o = attrn.clone().to(dev).requires_grad_(True), attrn.clone().requires_grad_(True)
        rs, rs_o = rast_o.clone().to(dev).requires_grad_(True), rast_o.clone().requires_grad_(True)
        out, da = raster.interpolate(attr, rs, T(f.astype(np.int32), dev), rast_db=db_o.to(dev), diff_attrs='all')
        out_o, da_o = OR.interpolate(attr_o, rs_o, tri_o, db_o)
        assert (out.detach().cpu() - out_o.detach()).abs().max() < 1e-5
        assert (da.cpu() - da_o.detach()).abs().max() < 1e-4 * max(1.0, da_o.abs().max().item())
        G = torch.randn(out_o.shape, generator=gen)
        (out * G.to(dev)).sum().backward()
        (out_o * G).sum().backward()
        assert (attr.grad.cpu() - attr_o.grad).abs().max() < 1e-4 * attr_o.grad.abs().max()
        assert (rs.grad.cpu()[..., :2] - rs_o.grad[..., :2]).abs().max() < 1e-4 * rs_o.grad.abs().max()
    # per-face attributes (render/render.py:264-267: face normals with index (f, f, f)) and a 1-channel attribute
    fa = torch.randn(1, f.shape[0], 3, generator=gen)
    fidx = np.repeat(np.arange(f.shape[0], dtype=np.int32)[:, None], 3, 1)
    o1, _ = raster.interpolate(fa.to(dev), rast_o.to(dev), T(fidx, dev))
    o2, _ = OR.interpolate(fa, rast_o, torch.from_numpy(fidx.astype(np.int64)))
    assert (o1.cpu() - o2).abs().max() < 1e-6


def check_gbuffer(dev, res=40):
    """d3h.raster.gbuffer (one pass: attribute groups + per-face gather + mask) against the oracle's dr.interpolate per attribute
    (render/render.py:257-267,283,328) and rast[..., 3] > 0: values, d(attr), d(face attr), d(rast); skipped groups; broadcast batch"""
    from d3h import raster
    from oracle import raster as OR
    posn, f = _raster_scene(res, 2)
    V, Fn = posn.shape[1], f.shape[0]
    gen = torch.Generator().manual_seed(16)
    tri_o = torch.from_numpy(f)
    tri = T(f.astype(np.int32), dev)
    rast_o, _ = OR.rasterize(torch.from_numpy(posn), tri_o, res, res)
    fidx_o = torch.arange(Fn)[:, None].expand(-1, 3).contiguous()
    widths = (3, 3, 3, 1)
    for batched, need in ((True, (True, True, True, True)), (False, (False, True, False, True)), (True, (True, False, True, False))):
        nb = 2 if batched else 1
        attrn = torch.randn(nb, V, 10, generator=gen)
        facen = torch.randn(2, Fn, 3, generator=gen)
        attr, attr_o = attrn.clone().to(dev).requires_grad_(True), attrn.clone().requires_grad_(True)
        fa, fa_o = facen.clone().to(dev).requires_grad_(True), facen.clone().requires_grad_(True)
        rs, rs_o = rast_o.clone().to(dev).requires_grad_(True), rast_o.clone().requires_grad_(True)
        groups, face_img, mask = raster.gbuffer(attr, widths, rs, tri, need=need, face_attr=fa, want_mask=True)
        out_o, _ = OR.interpolate(attr_o, rs_o, tri_o, None)
        face_o, _ = OR.interpolate(fa_o, rs_o, fidx_o, None)
        assert torch.equal(mask.cpu(), (rast_o[..., 3:] > 0).float())
        loss, loss_o, c0 = 0.0, 0.0, 0
        for k, w in enumerate(widths):
            if need[k]:
                assert groups[k].is_contiguous() and (groups[k].detach().cpu() - out_o[..., c0:c0 + w].detach()).abs().max() < 1e-5
                G = torch.randn(out_o[..., c0:c0 + w].shape, generator=gen)
                loss = loss + (groups[k] * G.to(dev)).sum()
                loss_o = loss_o + (out_o[..., c0:c0 + w] * G).sum()
            else:
                assert groups[k] is None
            c0 += w
        assert (face_img.detach().cpu() - face_o.detach()).abs().max() < 1e-6
        G = torch.randn(face_o.shape, generator=gen)
        (loss + (face_img * G.to(dev)).sum()).backward()
        (loss_o + (face_o * G).sum()).backward()
        assert (attr.grad.cpu() - attr_o.grad).abs().max() < 1e-4 * attr_o.grad.abs().max()
        assert (fa.grad.cpu() - fa_o.grad).abs().max() < 1e-4 * fa_o.grad.abs().max()
        assert (rs.grad.cpu()[..., :2] - rs_o.grad[..., :2]).abs().max() < 1e-4 * rs_o.grad.abs().max()
    # the rasteriser's backward folded into the G-buffer's (raster_pos=): same values, same d(attr) / d(face attr), and d(clip positions) equal to
    # rasterize-backward of the separate path's d(rast) (up to the order of the float atomics); broadcast and batched attributes
    for nb in (2, 1):
        attrn = torch.randn(nb, V, 10, generator=gen)
        facen = torch.randn(2, Fn, 3, generator=gen)
        Gs = [torch.randn(2, res, res, w, generator=gen).to(dev) for w in widths] + [torch.randn(2, res, res, 3, generator=gen).to(dev)]
        got = []
        for fold in (False, True):
            attr = attrn.clone().to(dev).requires_grad_(True)
            fa = facen.clone().to(dev).requires_grad_(True)
            pos = T(posn, dev, True)
            rs, _ = raster.rasterize(pos, tri, (res, res))
            assert rs.requires_grad
            groups, face_img, mask = raster.gbuffer(attr, widths, rs, tri, face_attr=fa, want_mask=True, raster_pos=pos if fold else None)
            # (nb == 2: every group takes a gradient -- 10 channels; nb == 1: two of the four groups only, as a tick that reads few buffers)
            use = range(4) if nb == 2 else (1, 3)
            loss = sum((groups[k] * Gs[k]).sum() for k in use) + (face_img * Gs[4]).sum()
            loss.backward()
            got.append(([g_.detach().clone() for g_ in groups] + [face_img.detach().clone()], attr.grad, fa.grad, pos.grad))
        for a_, b_ in zip(got[0][0], got[1][0]):
            assert torch.equal(a_, b_)
        assert float(got[0][3].abs().max()) > 0
        for k in (1, 2, 3):
            assert (got[0][k] - got[1][k]).abs().max() <= 2e-6 * got[0][k].abs().max(), k
        assert float(got[1][3][..., 2].abs().max()) == 0.0            # (no gradient into clip z: the barycentrics do not depend on it)
    # no face attribute, no mask, a single group
    g1, fi, m = raster.gbuffer(T(attrn[:, :, :4].numpy(), dev), (4,), rast_o.to(dev), tri, want_mask=False)
    o1, _ = OR.interpolate(attrn[:, :, :4], rast_o, tri_o, None)
    assert fi is None and m is None and (g1[0].cpu() - o1).abs().max() < 1e-5


def check_antialias(dev, res=40):
    from d3h import raster
    from oracle import raster as OR
    posn, f = _raster_scene(res, 2)
    tri_o = torch.from_numpy(f)
    rast_o, _ = OR.rasterize(torch.from_numpy(posn), tri_o, res, res)
    gen = torch.Generator().manual_seed(7)
    coln = torch.rand(2, res, res, 3, generator=gen)
    col, col_o = coln.clone().to(dev).requires_grad_(True), coln.clone().requires_grad_(True)
    pos, pos_o = T(posn, dev, True), torch.from_numpy(posn).requires_grad_(True)
    out = raster.antialias(col, rast_o.to(dev), pos, T(f.astype(np.int32), dev))
    out_o = OR.antialias(col_o, rast_o, pos_o, tri_o)
    assert (out_o.detach() - coln).abs().max() > 1e-3          # something was blended
    assert (out.detach().cpu() - out_o.detach()).abs().max() < 1e-4
    G = torch.randn(out_o.shape, generator=gen)
    (out * G.to(dev)).sum().backward()
    (out_o * G).sum().backward()
    assert (col.grad.cpu() - col_o.grad).abs().max() < 1e-4
    assert (pos.grad.cpu() - pos_o.grad).abs().max() < 2e-3 * pos_o.grad.abs().max()
    assert pos_o.grad.abs().max() > 0
    # silhouette edges EXACTLY on the midpoint between two pixel centres (d == 0.5: the blend weight |d - 0.5| is zero and its
    # derivative is 0 there, torch's abs'), with a 1/eps-sized upstream gradient on the untouched outer pixels -- what
    # F.normalize / cosine_similarity hand back for the exactly-zero background normals next to a silhouette
    R = 32
    quad = np.array([[[0.0, -0.5, 0.1, 1.0], [0.0, 0.5, 0.1, 1.0], [-0.5, 0.5, 0.1, 1.0], [-0.5, -0.5, 0.1, 1.0], [0.40, -0.123, 0.2, 1.0]]], np.float32)
    fq = np.array([[0, 1, 2], [0, 2, 3], [0, 4, 1]], np.int64)           # third triangle: a generic edge as well (non-zero gradient)
    rq, _ = OR.rasterize(torch.from_numpy(quad), torch.from_numpy(fq), R, R)
    cq = torch.rand(1, R, R, 3, generator=gen)
    Gq = torch.randn(1, R, R, 3, generator=gen)
    Gq[:, :, R // 4 - 1:R // 4 + 1, :] *= 1e13                             # the two pixel columns on either side of the x = -0.5 edge
    ca, co = cq.clone().to(dev).requires_grad_(True), cq.clone().requires_grad_(True)
    pa, po = T(quad, dev, True), torch.from_numpy(quad).requires_grad_(True)
    oa = raster.antialias(ca, rq.to(dev), pa, T(fq.astype(np.int32), dev))
    oo = OR.antialias(co, rq, po, torch.from_numpy(fq))
    assert (oa.detach().cpu() - oo.detach()).abs().max() < 1e-5
    (oa * Gq.to(dev)).sum().backward()
    (oo * Gq).sum().backward()
    assert 0 < float(po.grad.abs().max()) < 1e6, float(po.grad.abs().max())      # the oracle does not leak the spike ...
    assert (pa.grad.cpu() - po.grad).abs().max() < 2e-3 * po.grad.abs().max(), (pa.grad.cpu(), po.grad)       # ... and neither may the kernel


def check_texture(dev):
    from d3h import raster
    from oracle import raster as OR
    gen = torch.Generator().manual_seed(8)
    for tb in (1, 2):
        texn = torch.rand(tb, 16, 12, 3, generator=gen)
        uv = torch.rand(2, 10, 9, 2, generator=gen) * 1.2 - 0.1
        tex, tex_o = texn.clone().to(dev).requires_grad_(True), texn.clone().requires_grad_(True)
        out = raster.texture(tex, uv.to(dev), filter_mode='linear', boundary_mode='clamp')
        out_o = OR.texture(tex_o, uv)
        assert (out.detach().cpu() - out_o.detach()).abs().max() < 1e-6
        G = torch.randn(out_o.shape, generator=gen)
        (out * G.to(dev)).sum().backward()
        (out_o * G).sum().backward()
        assert (tex.grad.cpu() - tex_o.grad).abs().max() < 1e-5


# ---- mesh normals / shading normal / image loss / ssim / sdf_reg -------------------------------------------------
def _rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)


def check_normals_golden(dev):
    from d3h import imgops
    from oracle import image_ops as OI
    g = golden('imgops.npz')
    v = T(g['an_v'], dev, True)
    f32 = T(g['an_f'].astype(np.int32), dev)
    vn = imgops.auto_normals(v, f32)
    assert np.abs(vn.detach().cpu().numpy() - g['an_out']).max() < 2e-6
    (vn * T(g['an_g'], dev)).sum().backward()
    assert _rel(v.grad, g['an_dv']) < 1e-4
    # face normals vs the oracle restatement of render/render.py:261-264
    v2 = T(g['an_v'], dev, True)
    fn = imgops.face_normals(v2, f32)
    vo = torch.from_numpy(g['an_v']).requires_grad_(True)
    fo = OI.face_normals(vo, torch.from_numpy(g['an_f']))
    assert (fn.detach().cpu() - fo.detach()).abs().max() < 2e-5      # sliver triangles: cancellation in the cross product
    G = torch.randn(fo.shape, generator=torch.Generator().manual_seed(1))
    (fn * G.to(dev)).sum().backward()
    (fo * G).sum().backward()
    assert _rel(v2.grad, vo.grad.numpy()) < 1e-4


def check_shading_normal_golden(dev):
    from d3h import imgops
    g = golden('imgops.npz')
    for tag, two_sided in (('psn2_', True), ('psn1_', False)):
        t = {k: T(g[tag + 'in_' + k], dev, True) for k in ('pos', 'view', 'pert', 'snrm', 'stng', 'gnrm')}
        o = imgops.prepare_shading_normal(t['pos'], t['view'], t['pert'], t['snrm'], t['stng'], t['gnrm'], two_sided_shading=two_sided, opengl=True)
        assert np.abs(o.detach().cpu().numpy() - g[tag + 'out']).max() < 1e-5    # dot/0.1 amplifies fp32 rounding x10
        (o * T(g[tag + 'g'], dev)).sum().backward()
        for k, x in t.items():
            assert x.grad.shape == x.shape
            assert _rel(x.grad, g[tag + 'd_' + k]) < 2e-4, (tag, k)
    # default perturbed normal (None -> (0,0,1)), as render/render.py:111 calls it
    t = {k: T(g['psn2_in_' + k], dev) for k in ('pos', 'view', 'snrm', 'stng', 'gnrm')}
    o = imgops.prepare_shading_normal(t['pos'], t['view'], None, t['snrm'], t['stng'], t['gnrm'])
    assert torch.isfinite(o).all()


def check_image_loss_golden(dev):
    from d3h import imgops
    from oracle import image_ops as OI
    g = golden('imgops.npz')
    for loss in ('l1', 'mse', 'smape', 'relmse'):
        a, b = T(g['il_a'], dev, True), T(g['il_b'], dev, True)
        l = imgops.image_loss(a, b, loss=loss, tonemapper='none')
        assert abs(l.item() - float(g[f'il_{loss}'])) < 1e-6 * max(1.0, abs(float(g[f'il_{loss}'])))
        l.backward()
        assert _rel(a.grad, g[f'il_{loss}_da']) < 1e-4 and _rel(b.grad, g[f'il_{loss}_db']) < 1e-4
        # log_srgb: the live CUDA semantics (loss.cu), restated by the oracle
        a2, b2 = T(g['il_a'], dev, True), T(g['il_b'], dev, True)
        l2 = imgops.image_loss(a2, b2, loss=loss, tonemapper='log_srgb')
        ao, bo = torch.from_numpy(g['il_a']).requires_grad_(True), torch.from_numpy(g['il_b']).requires_grad_(True)
        lo = OI.image_loss(ao, bo, loss, 'log_srgb')
        assert abs(l2.item() - lo.item()) < 2e-6
        l2.backward(); lo.backward()
        assert _rel(a2.grad, ao.grad.numpy()) < 1e-3 and _rel(b2.grad, bo.grad.numpy()) < 1e-3


def check_xfm_points(dev, n=777):
    """render.renderutils.xfm_points on the one-thread-per-point kernel against the reference's matmul formulation
    (render/renderutils/ops.py:518-537): batched and broadcast points, values and d(points)"""
    import render.renderutils as ru
    gen = torch.Generator().manual_seed(21)
    M = torch.randn(3, 4, 4, generator=gen)
    for nbp in (3, 1):
        pn = torch.randn(nbp, n, 3, generator=gen)
        p, p_o = pn.clone().to(dev).requires_grad_(True), pn.clone().requires_grad_(True)
        out = ru.xfm_points(p, M.to(dev))
        ref = torch.matmul(torch.nn.functional.pad(p_o, pad=(0, 1), mode='constant', value=1.0), torch.transpose(M, 1, 2))
        assert out.shape == ref.shape and (out.detach().cpu() - ref.detach()).abs().max() < 2e-6 * ref.abs().max()
        G = torch.randn(ref.shape, generator=gen)
        (out * G.to(dev)).sum().backward()
        (ref * G).sum().backward()
        assert (p.grad.cpu() - p_o.grad).abs().max() < 2e-6 * p_o.grad.abs().max()
    assert ru.xfm_points(torch.zeros(1, 0, 3, device=dev), M.to(dev)).shape == (3, 0, 4)
    # a trainable matrix keeps the matmul formulation
    Mg = M.clone().to(dev).requires_grad_(True)
    ru.xfm_points(torch.randn(3, 5, 3, generator=gen).to(dev), Mg).sum().backward()
    assert Mg.grad is not None


def check_sample_points(dev, nv=300, nf=500, n=4000):
    """kaolin shim.  Fused sampler (default): every sample lies on the triangle it reports, zero-area rows (degenerate faces, the zero padding of a
    face list at its allocation bound) are never picked, the pick frequencies follow the face areas, the barycentric map is the reference
    formula on the recorded random numbers.  torch.multinomial form (D3H_FUSED_SAMPLER=0): equal to its differentiable torch formulation on
    the same random stream."""
    import kaolin.ops.mesh as K
    gen = torch.Generator().manual_seed(2)
    v = torch.randn(nv, 3, generator=gen).to(dev)
    f = torch.randint(0, nv, (nf, 3), generator=gen).to(dev)
    f[::7] = f[::7][:, :1].expand(-1, 3)                 # every 7th face degenerate (a, a, a)
    f = torch.cat([f, torch.zeros(nf // 3, 3, dtype=f.dtype, device=f.device)])      # + zero padding rows
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    areas = 0.5 * torch.linalg.norm(torch.cross(b - a, c - a, dim=-1), dim=-1)
    assert K.FUSED_SAMPLER
    torch.manual_seed(7)
    with torch.no_grad():
        p0, i0 = K.sample_points(v[None], f, n)
    assert p0.shape == (1, n, 3) and i0.shape == (1, n) and i0.dtype == torch.int64
    pick = i0[0]
    assert int(pick.min()) >= 0 and int(pick.max()) < f.shape[0]
    assert bool((areas[pick] > 0).all()), 'a zero-area face was picked'
    torch.manual_seed(7)
    rnd = torch.rand(n, 3, device=v.device)              # the sampler's one draw
    u, w = rnd[:, 1:2].sqrt(), rnd[:, 2:3]
    ref = (1 - u) * a[pick] + u * (1 - w) * b[pick] + u * w * c[pick]
    assert (p0[0] - ref).abs().max() < 1e-5
    # the pick is the inverse CDF of the areas at rnd[:, 0]  (float64 reference; a sample within 1e-6 of a step may fall on either side)
    cdf = torch.cumsum(areas.double(), 0)
    want = torch.searchsorted(cdf, rnd[:, 0].double() * cdf[-1], right=True).clamp(max=f.shape[0] - 1)
    off = pick != want
    # (float32 prefix sums: a sample within ~1e-7 of the total area of a step may fall on either side of it -- n * rows * 2.4e-7 of them expected)
    assert int(off.sum()) <= max(2, n // 1000, int(n * f.shape[0] * 1e-6)), int(off.sum())
    r = rnd[:, 0].double() * cdf[-1]
    assert bool(((cdf[pick[off]] - r[off]).abs().minimum((cdf[(pick[off] - 1).clamp(min=0)] - r[off]).abs()) < 1e-5 * cdf[-1]).all())
    # frequencies ~ areas: 20 x n samples into 8 area-sorted bins
    torch.manual_seed(8)
    with torch.no_grad():
        _, big = K.sample_points(v[None], f, 20 * n)
    order = torch.argsort(areas)
    bins = torch.chunk(order, 8)
    cnt = torch.bincount(big[0], minlength=f.shape[0]).double()
    for bn in bins:
        pexp = float(areas[bn].sum() / areas.sum())
        got = float(cnt[bn].sum()) / (20 * n)
        assert abs(got - pexp) <= 5 * (pexp * (1 - pexp) / (20 * n)) ** 0.5 + 1e-4, (got, pexp)
    # a mesh without any area: defined output (the caller discards it), no fault
    z = torch.zeros(5, 3, dtype=f.dtype, device=f.device)
    with torch.no_grad():
        pz, iz = K.sample_points(v[None], z, 64)
    assert bool(torch.isfinite(pz).all()) and int(iz.max()) < 5
    # ---- the torch.multinomial form -----------------------------------------------------------------------------------------------
    K.FUSED_SAMPLER = False
    try:
        torch.manual_seed(7)
        with torch.no_grad():
            p0, i0 = K.sample_points(v[None], f, n)
        torch.manual_seed(7)
        pick = torch.multinomial(K._pick_weights(areas), n, replacement=True)
        uw = torch.rand(n, 2, device=v.device)
        u, w = uw[:, :1].sqrt(), uw[:, 1:]
        ref = (1 - u) * a[pick] + u * (1 - w) * b[pick] + u * w * c[pick]
        # the kernel's areas and torch's differ in the last bit, and torch.multinomial's normalisation sums with atomics: a sample within float32
        # rounding of a step of the cumulative distribution may land on either side of it -- ~ n * rows * 1e-7 of them (at n = 50 000 and
        # 80 000 rows: within the round-5 bar of n // 1000 = 50 on that round's boxes, 55 on a round-6 box)
        same = i0[0] == pick
        assert int((~same).sum()) <= max(n // 1000, int(n * f.shape[0] * 1e-7)), int((~same).sum())
        assert (p0[0] - ref)[same].abs().max() < 1e-5
    finally:
        K.FUSED_SAMPLER = True
    vg = v.clone().requires_grad_(True)
    p1, i1 = K.sample_points(vg[None], f, n)              # differentiable path still there
    assert p1.requires_grad and p1.shape == (1, n, 3)


def check_first_channels(dev, B=2, H=11, W=7):
    """d3h.imgops.first_channels: the values of x[..., :k] (a view, no copy) and a gradient equal to autograd's slice node's, for a contiguous and a
    strided upstream gradient; plain slicing when no gradient is wanted"""
    from d3h import imgops as I
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, 6, generator=gen).to(dev)
    w = torch.randn(B, H, W, 3, generator=gen).to(dev)
    for strided in (False, True):
        a = x.clone().requires_grad_(True)
        b = x.clone().requires_grad_(True)
        ya, yb = I.first_channels(a, 3), b[..., :3]
        assert ya.shape == yb.shape and torch.equal(ya, yb) and ya.data_ptr() == a.data_ptr()
        if strided:
            (ya.permute(0, 3, 1, 2) * w.permute(0, 3, 1, 2)).sum().backward()
            (yb.permute(0, 3, 1, 2) * w.permute(0, 3, 1, 2)).sum().backward()
        else:
            (ya * w).sum().backward()
            (yb * w).sum().backward()
        assert torch.equal(a.grad, b.grad) and bool((a.grad[..., 3:] == 0).all())
    with torch.no_grad():
        assert torch.equal(I.first_channels(x, 3), x[..., :3])


def check_composite(dev, B=2, H=13, W=17):
    """fused composite vs the reference's per-buffer formulation (render.py:375-382,430-449): lerp(bg, [values, 1], coverage * alpha)"""
    from d3h import imgops as I
    gen = torch.Generator().manual_seed(5)
    rast = torch.zeros(B, H, W, 4)
    rast[..., 3] = (torch.rand(B, H, W, generator=gen) > 0.4).float() * torch.randint(1, 50, (B, H, W), generator=gen).float()
    wide = torch.randn(B, H, W, 6, generator=gen)
    gn = torch.randn(B, H, W, 3, generator=gen)
    packed = torch.rand(B, H, W, 10, generator=gen) * 2 - 1
    depth = torch.rand(B, H, W, 1, generator=gen) * 3
    bg = torch.rand(B, H, W, 3, generator=gen)
    cov = (rast[..., 3:] > 0).float()

    def ref(wide, gn, packed, depth):
        outs = []
        for vals, bgk in ((wide[..., 0:3], torch.cat((bg, torch.zeros_like(bg[..., :1])), -1)), (gn, None), (wide[..., 3:6], None), (depth, 20.0)):
            buf = torch.cat((vals, torch.ones_like(vals[..., :1])), -1)
            a = cov * buf[..., -1:]
            b_ = torch.zeros_like(buf) if bgk is None else (torch.full_like(buf, bgk) if isinstance(bgk, float) else bgk)
            outs.append(torch.lerp(b_, torch.cat((buf[..., :-1], torch.ones_like(buf[..., -1:])), -1), a))
        m = packed[..., 9:10]
        outs.append(torch.lerp(torch.zeros_like(m), torch.ones_like(m), cov * m))
        return torch.cat(outs, -1)

    def ours(wide, gn, packed, depth):
        return I.composite(rast.to(dev), [(wide[..., 0:3], I.COMP_IMAGE, bg.to(dev)), (gn, I.COMP_ZERO, None), (wide[..., 3:6], I.COMP_ZERO, None),
                                          (depth, I.COMP_CONST20, None), (packed[..., 9:10], I.COMP_ALPHA, None)])

    w = torch.randn(B, H, W, 15, generator=gen)
    a_in = [t.clone().requires_grad_(True) for t in (wide, gn, packed, depth)]
    b_in = [t.clone().to(dev).requires_grad_(True) for t in (wide, gn, packed, depth)]
    r = ref(*a_in)
    o = ours(*b_in)
    assert o.shape == r.shape
    assert (o.detach().cpu() - r.detach()).abs().max() <= 1.2e-7       # exact except lerp's rounding of coverage * msdf
    (r * w).sum().backward()
    (o * w.to(dev)).sum().backward()
    for x, y in zip(a_in, b_in):
        assert torch.equal(x.grad, y.grad.cpu())
    # single shared background image and a broadcast source
    o2 = I.composite(rast.to(dev), [(wide[..., 0:3].to(dev), I.COMP_IMAGE, bg[:1].to(dev)), (gn[:1].to(dev), I.COMP_ZERO, None)])
    exp = torch.where(cov > 0, torch.cat((wide[..., 0:3], torch.ones(B, H, W, 1)), -1), torch.cat((bg[:1].expand(B, -1, -1, -1), torch.zeros(B, H, W, 1)), -1))
    assert torch.equal(o2[..., 0:4].cpu(), exp)
    assert torch.equal(o2[..., 4:7].cpu(), gn[:1].expand(B, -1, -1, -1) * cov)


def check_composite_antialias_fused(dev, res=40):
    """the forward-only fused composite + antialias kernel == antialias(composite(...)) bit for bit, on a two-frame raster scene with every
    source kind (image background batched / shared, constant 20, alpha-only, zero background, a strided slice as source)"""
    from d3h import imgops as I, raster
    from oracle import raster as OR
    posn, f = _raster_scene(res, 2)
    tri_o = torch.from_numpy(f)
    rast_o, _ = OR.rasterize(torch.from_numpy(posn), tri_o, res, res)
    rast = rast_o.to(dev).contiguous()
    pos, tri = T(posn, dev), T(f.astype(np.int32), dev)
    gen = torch.Generator().manual_seed(9)
    B, H, W = 2, res, res
    wide = torch.randn(B, H, W, 6, generator=gen).to(dev)
    for bg_b in (1, B):
        sources = [(wide[..., 0:3], I.COMP_IMAGE, torch.rand(bg_b, H, W, 3, generator=gen).to(dev)),
                   (torch.randn(B, H, W, 3, generator=gen).to(dev), I.COMP_ZERO, None),
                   (torch.rand(B, H, W, 1, generator=gen).to(dev) * 3, I.COMP_CONST20, None),
                   (torch.rand(B, H, W, 1, generator=gen).to(dev) * 2 - 1, I.COMP_ALPHA, None),
                   (wide[..., 3:6], I.COMP_ZERO, None),
                   (torch.randn(1, H, W, 2, generator=gen).to(dev), I.COMP_ZERO, None)]          # a [1,...] source broadcast over the batch
        with torch.no_grad():
            ref = raster.antialias(I.composite(rast, sources), rast, pos, tri)
            got = I.composite_antialias(rast, sources, pos, tri)
        assert got.shape == ref.shape and got.shape[-1] == 4 + 4 + 2 + 1 + 4 + 3
        assert (ref != I.composite(rast, sources)).any(), 'the scene has no antialiased pixel'
        assert torch.equal(got, ref), float((got - ref).abs().max())
        # the differentiable fused pair (one kernel each way) against the two separate ops with their two backwards: values and source
        # gradients bit for bit (sources without a gradient request and the broadcast source included), d(pos) up to the order of its atomics
        wgt = torch.randn(ref.shape, generator=gen).to(dev)
        wgt[..., 8:10] = 0                                        # a buffer nobody reads: zero upstream gradient (skipped products)
        grads = []
        for fused in (False, True):
            leaf = [s_.clone().requires_grad_(k != 2) for k, (s_, _, _) in enumerate(sources)]      # (source 2 takes no gradient)
            srcs_l = [(t, kind, bg) for t, (_, kind, bg) in zip(leaf, sources)]
            p_l = pos.clone().requires_grad_(True)
            o = I.composite_antialias_grad(rast, srcs_l, p_l, tri) if fused else raster.antialias(I.composite(rast, srcs_l), rast, p_l, tri)
            assert torch.equal(o.detach(), ref)
            (o * wgt).sum().backward()
            grads.append(([t.grad for t in leaf], p_l.grad))
        for a_, b_ in zip(*[g[0] for g in grads]):
            assert (a_ is None) == (b_ is None)
            if a_ is not None:
                assert torch.equal(a_, b_), float((a_ - b_).abs().max())
        assert grads[0][0][2] is None and grads[0][1].abs().max() > 0
        assert (grads[0][1] - grads[1][1]).abs().max() <= 1e-5 * grads[0][1].abs().max()
        # a position-only request (no source takes a gradient) and a source-only request
        p_l = pos.clone().requires_grad_(True)
        o = I.composite_antialias_grad(rast, sources, p_l, tri)
        (o * wgt).sum().backward()
        assert (p_l.grad - grads[0][1]).abs().max() <= 1e-5 * grads[0][1].abs().max()
        leaf = [s_.clone().requires_grad_(True) for s_, _, _ in sources]
        o = I.composite_antialias_grad(rast, [(t, kind, bg) for t, (_, kind, bg) in zip(leaf, sources)], pos, tri)
        (o * wgt).sum().backward()
        assert torch.equal(leaf[0].grad, grads[0][0][0]) and leaf[2].grad is not None
        # a source given as "the first 3 channels of a 6-channel tensor" (the shaded colour inside the texture MLP's output): same values,
        # the gradient at the tensor's width with zeros behind the prefix
        wl = wide.clone().requires_grad_(True)
        o = I.composite_antialias_grad(rast, [(wl, I.COMP_IMAGE, sources[0][2], 3)] + list(sources[1:4]), pos, tri)
        wr = wide.clone().requires_grad_(True)
        r_ = I.composite_antialias_grad(rast, [(wr[..., 0:3], I.COMP_IMAGE, sources[0][2])] + list(sources[1:4]), pos, tri)
        assert torch.equal(o.detach(), r_.detach())
        (o * wgt[..., :o.shape[-1]]).sum().backward()
        (r_ * wgt[..., :o.shape[-1]]).sum().backward()
        assert torch.equal(wl.grad, wr.grad) and float(wl.grad[..., 3:].abs().max()) == 0.0 and float(wl.grad[..., :3].abs().max()) > 0
    # an empty mesh (the body pass of a fresh split stage can extract nothing): every pixel uncovered, backgrounds only
    empty_tri = torch.zeros(0, 3, dtype=torch.int32, device=dev)
    rast0 = torch.zeros_like(rast)
    with torch.no_grad():
        got0 = I.composite_antialias(rast0, sources, pos[:, :0].contiguous(), empty_tri)
        ref0 = I.composite(rast0, sources)
    assert torch.equal(got0, ref0)
    leaf = [s_.clone().requires_grad_(True) for s_, _, _ in sources]
    o0 = I.composite_antialias_grad(rast0, [(t, kind, bg) for t, (_, kind, bg) in zip(leaf, sources)], pos[:, :0].contiguous(), empty_tri)
    assert torch.equal(o0.detach(), ref0)
    o0.sum().backward()
    assert all(float(t.grad.abs().max()) == 0.0 for t in leaf)          # nothing is covered: no source receives anything


def check_material_grads(dev, B=2, H=19, W=23):
    """fused kd / kd_grad / ks_grad / nrm_grad of shade() vs the reference's own lines (render.py:72-74,88-91,104-105) as torch ops: values and
    all four input gradients bit-equal (abs, subtract, one multiply: no rounding freedom), ties (|0|) included; the partial forms too"""
    from d3h import imgops as I
    gen = torch.Generator().manual_seed(11)
    tex, texj = torch.rand(B, H, W, 6, generator=gen), torch.rand(B, H, W, 6, generator=gen)
    texj[0, :3] = tex[0, :3]                                  # exact ties: d|x|/dx = 0 there
    nrm, nrmj = torch.randn(B, H, W, 3, generator=gen), torch.randn(B, H, W, 3, generator=gen)
    nrmj[1, 2:5] = nrm[1, 2:5]
    mask = (torch.rand(B, H, W, 1, generator=gen) > 0.3).float()
    mask_tap = torch.rand(B, H, W, 1, generator=gen)
    ws = [torch.randn(B, H, W, 3, generator=gen) for _ in range(4)]

    def ref(tex, texj, nrm, nrmj):
        kd, ks = tex[..., 0:3], tex[..., 3:6]
        kd_grad = torch.abs(texj[..., 0:3] - kd)
        ks_grad = torch.abs(texj[..., 3:6] - ks) * torch.tensor([0, 1, 1], dtype=torch.float32)[None, None, None, :]
        nrm_grad = torch.abs(nrmj - nrm) * (mask * mask_tap)
        return kd, kd_grad, ks_grad, nrm_grad

    a_in = [t.clone().requires_grad_(True) for t in (tex, texj, nrm, nrmj)]
    b_in = [t.clone().to(dev).requires_grad_(True) for t in (tex, texj, nrm, nrmj)]
    r = ref(*a_in)
    o = I.material_grads(b_in[0], b_in[1], b_in[2], b_in[3], mask.to(dev), mask_tap.to(dev))
    for x, y in zip(r, o):
        assert torch.equal(x.detach(), y.detach().cpu())
    sum((x * w).sum() for x, w in zip(r, ws)).backward()
    sum((y * w.to(dev)).sum() for y, w in zip(o, ws)).backward()
    for x, y in zip(a_in, b_in):
        assert torch.equal(x.grad, y.grad.cpu())
    # without the normal part, and with a gradient on two of the three outputs only
    t2, j2 = tex.clone().to(dev).requires_grad_(True), texj.clone().to(dev).requires_grad_(True)
    kd, kdg, ksg, ng = I.material_grads(t2, j2)
    assert ng is None and torch.equal(kd.detach().cpu(), tex[..., 0:3])
    ((kd * ws[0].to(dev)).sum() + (ksg * ws[2].to(dev)).sum()).backward()
    a2 = [t.clone().requires_grad_(True) for t in (tex, texj, nrm, nrmj)]
    r2 = ref(*a2)
    ((r2[0] * ws[0]).sum() + (r2[2] * ws[2]).sum()).backward()
    assert torch.equal(a2[0].grad, t2.grad.cpu()) and torch.equal(a2[1].grad, j2.grad.cpu())


def check_pixel_losses(dev, B=2, H=26, W=22, with_ssim=True):
    """fused per-pixel loss stack vs the torch composition tick_init / tick_split use (hmsdf.py:835-839,895-898), on the oracle's
    image_loss / ssim restatements; values and the gradient w.r.t. the stacked render output"""
    import torch.nn.functional as F
    from d3h import imgops
    from oracle import image_ops as O
    gen = torch.Generator().manual_seed(11)
    C = 23                                                   # [pad, shaded rgba, pad, gn xyz+a, msdf, kd_grad, ks_grad, normal_grad]
    layout = {'shaded': (1, 4), 'geometric_normal': (6, 4), 'msdf_image': (10, 1), 'kd_grad': (11, 4), 'ks_grad': (15, 4), 'normal_grad': (19, 4)}
    st = torch.rand(B, H, W, C, generator=gen) * 1.4 - 0.2
    st[..., 10] = torch.rand(B, H, W, generator=gen) * 2 - 1
    st[0, :4, :, 6:9] = 0.0                                  # background: zero geometric normal (normalize's eps branch)
    cref = torch.rand(B, H, W, 4, generator=gen)
    a = torch.rand(B, H, W, generator=gen)
    cref[..., 3] = torch.where(a < 0.4, torch.zeros_like(a), torch.where(a < 0.8, torch.ones_like(a), a))
    nref = torch.randn(B, H, W, 3, generator=gen)
    nref[0, :6] = 0.0                                        # zero reference normals over (part of) the zero-normal background
    w = torch.tensor([100.0, 1.0, 0.5, 0.5, 1.0, -0.1, 0.1, 0.05, 0.025, -1.0])

    def torch_side(x):
        sh, gn, mi = x[..., 1:5], x[..., 6:10], x[..., 10:11]
        gt_mask = cref[..., 3:]
        v = [F.mse_loss(sh[..., 3:], cref[..., 3:]),
             O.image_loss(sh[..., 0:3] * cref[..., 3:], cref[..., 0:3] * cref[..., 3:], 'l1', 'log_srgb'),
             F.l1_loss(mi.clamp(min=0) * (gt_mask == 0).float(), torch.zeros_like(gt_mask)),
             F.l1_loss(mi.clamp(max=0) * (gt_mask == 1).float(), torch.ones_like(gt_mask))]
        out_n = F.normalize(gn[..., 0:3], p=2, dim=-1) * torch.tensor([1.0, -1.0, -1.0])
        gt_n = F.normalize(nref, p=2, dim=-1)
        v.append(F.mse_loss(out_n, gt_n))
        v.append(F.cosine_similarity(out_n.reshape(-1, 3), gt_n.reshape(-1, 3), dim=1).mean())
        kg, sg, ng = x[..., 11:15], x[..., 15:19], x[..., 19:23]                # regularizer.material_smoothness_grad, term by term
        v.append(torch.mean((kg[..., 0] + kg[..., 1] + kg[..., 2]) / 3 * kg[..., -1]))
        v.append(torch.mean(sg[..., :-1] * sg[..., -1:]))
        v.append(torch.mean(ng[..., :-1] * ng[..., -1:]))
        if with_ssim:
            v.append(O.ssim((sh[..., 0:3] * cref[..., 3:]).permute(0, 3, 1, 2), (cref[..., 0:3] * cref[..., 3:]).permute(0, 3, 1, 2)))
        else:
            v.append(torch.zeros(()))
        return torch.stack(v)

    x0 = st.clone().requires_grad_(True)
    ref = torch_side(x0)
    (ref * w).sum().backward()
    x1 = st.clone().to(dev).requires_grad_(True)
    d = imgops.pixel_losses(x1, layout, cref.to(dev), nref.to(dev), ('l1', 'log_srgb'), want_ssim=with_ssim)
    got = torch.stack([d[k] for k in imgops.PIXEL_LOSS_KEYS])
    (got * w.to(dev)).sum().backward()
    for k, a_, b_ in zip(imgops.PIXEL_LOSS_KEYS, got.detach().cpu().tolist(), ref.detach().tolist()):
        assert abs(a_ - b_) <= 2e-6 + 2e-5 * abs(b_), (k, a_, b_)
    gd, gr = x1.grad.cpu(), x0.grad
    assert gd[..., 0].abs().max() == 0 and gd[..., 5].abs().max() == 0 and gd[..., 9].abs().max() == 0
    # the zero-normal pixels carry torch's 1/eps gradients (1e12 and beyond): compare relatively, per channel group
    for sl in (slice(1, 5), slice(6, 9), slice(10, 11), slice(11, 15), slice(15, 19), slice(19, 23)):
        num = (gd[..., sl] - gr[..., sl]).abs()
        den = gr[..., sl].abs()
        assert bool((num <= 1e-4 * den + 1e-5 * den.max().clamp(max=1.0)).all()), (sl, float(num.max()), float(den.max()))
    # the masked colour image as a second output (the LPIPS input of tick_split): plain, and mapped like lpips.py's normalize + ScalingLayer;
    # its gradient re-enters the same backward pass
    shift, scale = [-.030, -.088, -.188], [.458, .448, .450]
    wm = torch.randn(B, H, W, 3, generator=gen)
    for prep in ((), (shift, scale)):
        x0 = st.clone().requires_grad_(True)
        m_ref = x0[..., 1:4] * cref[..., 3:]
        if prep:
            m_ref = ((2 * m_ref - 1) - torch.tensor(shift)) / torch.tensor(scale)
        ((torch_side(x0) * w).sum() + (m_ref * wm).sum()).backward()
        x1 = st.clone().to(dev).requires_grad_(True)
        d = imgops.pixel_losses(x1, layout, cref.to(dev), nref.to(dev), ('l1', 'log_srgb'), want_ssim=with_ssim, masked_prep=prep)
        assert torch.equal(d['masked'].detach().cpu(), m_ref.detach())
        ((d['vec'] * w.to(dev)).sum() + (d['masked'] * wm.to(dev)).sum()).backward()
        num, den = (x1.grad.cpu()[..., 1:5] - x0.grad[..., 1:5]).abs(), x0.grad[..., 1:5].abs()
        assert bool((num <= 1e-4 * den + 1e-5 * den.max().clamp(max=1.0)).all()), (prep, float(num.max()))
        # gradient through the image alone (the loss vector unused)
        x2 = st.clone().to(dev).requires_grad_(True)
        d2 = imgops.pixel_losses(x2, layout, cref.to(dev), nref.to(dev), ('l1', 'log_srgb'), want_ssim=False, masked_prep=prep)
        (d2['masked'] * wm.to(dev)).sum().backward()
        x3 = st.clone().requires_grad_(True)
        m3 = x3[..., 1:4] * cref[..., 3:]
        if prep:
            m3 = ((2 * m3 - 1) - torch.tensor(shift)) / torch.tensor(scale)
        (m3 * wm).sum().backward()
        assert (x2.grad.cpu() - x3.grad).abs().max() <= 1e-6 * x3.grad.abs().max()


def check_pixel_losses_ssim_occupancy(dev, B=3, H=150, W=330):
    """the occupancy cells of the SSIM operands (csrc/image_ops.hip: SsimOcc): with a target alpha that is zero outside a blob (a frame with the
    blob at the image border, one in the middle, one EMPTY frame) the bands that see only zeros are skipped -- the SSIM value and the
    gradient must be those of the run that computes every band (gradient bit for bit) and of the oracle's ssim"""
    from d3h import imgops
    from oracle import image_ops as O
    gen = torch.Generator().manual_seed(31)
    C = 5
    layout = {'shaded': (0, 4)}
    st = torch.rand(B, H, W, C, generator=gen)
    cref = torch.rand(B, H, W, 4, generator=gen)
    alpha = torch.zeros(B, H, W)
    alpha[0, 0:40, 250:330] = 1.0                                # touches the top-right corner (cells at the image border, ragged last column)
    alpha[0, 100:101, 3:4] = 0.5                                 # a single pixel far from the blob
    alpha[B - 1, 60:110, 120:200] = torch.rand(50, 80, generator=gen)
    cref[..., 3] = alpha                                         # frame 1 (of three): nothing
    res = {}
    for occ_on in (True, False):
        imgops.SSIM_OCC = occ_on
        try:
            x = st.clone().to(dev).requires_grad_(True)
            d = imgops.pixel_losses(x, layout, cref.to(dev), None, ('l1', 'log_srgb'), want_ssim=True)
            d['ssim'].backward()
            res[occ_on] = (float(d['ssim'].detach()), x.grad.cpu().clone())
        finally:
            imgops.SSIM_OCC = True
    x0 = st.clone().requires_grad_(True)
    ref = O.ssim((x0[..., 0:3] * cref[..., 3:]).permute(0, 3, 1, 2), (cref[..., 0:3] * cref[..., 3:]).permute(0, 3, 1, 2))
    ref.backward()
    assert abs(res[True][0] - res[False][0]) <= 2e-6 and abs(res[True][0] - float(ref)) <= 5e-6, (res[True][0], res[False][0], float(ref))
    assert torch.equal(res[True][1], res[False][1]), float((res[True][1] - res[False][1]).abs().max())
    assert (B < 3 or float(res[True][1][1].abs().max()) == 0.0) and float(res[True][1].abs().max()) > 0
    assert (res[True][1] - x0.grad).abs().max() <= 1e-4 * x0.grad.abs().max()


def check_seq_losses(dev, B=2, H=21, W=19):
    """fused mask / image terms of tick_seq vs the reference's lines (hmsdf.py:787-797,1110-1123) as torch ops on the oracle's image_loss
    restatement: the six means and the gradient w.r.t. the stacked render output"""
    import torch.nn.functional as F
    from d3h import imgops
    from oracle import image_ops as O
    gen = torch.Generator().manual_seed(23)
    C = 12
    layout = {'shaded': (1, 4), 'geometric_normal': (6, 4)}
    st = torch.rand(B, H, W, C, generator=gen) * 1.3 - 0.15
    st[..., 9] = torch.rand(B, H, W, generator=gen).clamp(0.05, 1.0)              # the antialiased coverage (no exact zeros here: see below)
    label = (torch.rand(B, H, W, generator=gen) > 0.5).float()
    gts = [torch.rand(B, H, W, 4, generator=gen) for _ in range(3)]
    for g_ in gts:
        g_[..., 3] = (g_[..., 3] > 0.4).float()
    w = torch.tensor([200.0, 200.0, 200.0, 1.0, 0.7, 1.3])
    for spec in (('l1', 'log_srgb'), ('mse', 'none')):
        def torch_side(x):
            alpha = x[..., 9]
            masks = [alpha[..., None], (label * alpha)[..., None], ((1 - label) * alpha)[..., None]]
            rgb = x[..., 1:4]
            v = [F.mse_loss(m, g_[..., 3:]) for m, g_ in zip(masks, gts)]
            v += [O.image_loss(rgb * m, g_[..., 0:3], spec[0], spec[1]) for m, g_ in zip(masks, gts)]
            return torch.stack(v)
        # (without a tone mapper the kernels -- like image_loss / pixel_losses -- pass the gradient through the [0, 65535] clamp of the
        # forward, as loss.cu:155-193 does; torch.clamp does not: negative colours only under the tone mapper, whose rule both sides share)
        st_ = st if spec[1] != 'none' else st.clamp(min=0.01)
        x0 = st_.clone().requires_grad_(True)
        ref = torch_side(x0)
        (ref * w).sum().backward()
        x1 = st_.clone().to(dev).requires_grad_(True)
        got = imgops.seq_losses(x1, layout, label.to(dev), *[g_.to(dev) for g_ in gts], spec)
        (got * w.to(dev)).sum().backward()
        for k, a_, b_ in zip(imgops.SEQ_LOSS_KEYS, got.detach().cpu().tolist(), ref.detach().tolist()):
            assert abs(a_ - b_) <= 2e-6 + 2e-5 * abs(b_), (spec, k, a_, b_)
        gd, gr = x1.grad.cpu(), x0.grad
        assert gd[..., 0].abs().max() == 0 and gd[..., 4:9].abs().max() == 0 and gd[..., 10:].abs().max() == 0
        num, den = (gd - gr).abs(), gr.abs()
        assert bool((num <= 1e-4 * den + 1e-6 * den.max()).all()), (spec, float(num.max()), float(den.max()))
    # uncovered pixels (coverage exactly 0: the masked colour is exactly 0): the reference's loss kernel passes no gradient AT the clamp bounds
    # (loss.cu:44-62), unlike torch.clamp -- compared with the composition tick_seq used before, built on the image-loss kernel itself
    st0 = st.clone()
    st0[0, :4, :, 9] = 0.0
    x0 = st0.clone().to(dev).requires_grad_(True)
    alpha = x0[..., 9]
    lab = label.to(dev)
    masks = [alpha[..., None], (lab * alpha)[..., None], ((1 - lab) * alpha)[..., None]]
    v = [F.mse_loss(m, g_.to(dev)[..., 3:]) for m, g_ in zip(masks, gts)]
    v += [imgops.image_loss(x0[..., 1:4] * m, g_.to(dev)[..., 0:3], 'l1', 'log_srgb') for m, g_ in zip(masks, gts)]
    (torch.stack(v) * w.to(dev)).sum().backward()
    x1 = st0.clone().to(dev).requires_grad_(True)
    (imgops.seq_losses(x1, layout, lab, *[g_.to(dev) for g_ in gts], ('l1', 'log_srgb')) * w.to(dev)).sum().backward()
    num, den = (x1.grad - x0.grad).abs().cpu(), x0.grad.abs().cpu()
    assert bool((num <= 1e-4 * den + 1e-6 * den.max()).all()), (float(num.max()), float(den.max()))


def check_ssim_golden(dev):
    from d3h import imgops
    g = golden('imgops.npz')
    x, y = T(g['ssim_x'], dev, True), T(g['ssim_y'], dev, True)
    s = imgops.ssim(x, y)
    assert abs(s.item() - float(g['ssim'])) < 2e-6
    s.backward()
    assert _rel(x.grad, g['ssim_dx']) < 2e-4 and _rel(y.grad, g['ssim_dy']) < 2e-4
    # a constant second image (the target of a loss): the three-plane form of the kernels (need_b = 0) -- same value, same d/dx
    x2, y2 = T(g['ssim_x'], dev, True), T(g['ssim_y'], dev)
    s2 = imgops.ssim(x2, y2)
    assert abs(s2.item() - float(g['ssim'])) < 2e-6
    s2.backward()
    assert _rel(x2.grad, g['ssim_dx']) < 2e-4


def check_sdf_reg_golden(dev):
    from d3h import imgops
    g = golden('imgops.npz')
    sdf = T(g['reg_sdf'], dev, True)
    e = T(g['reg_edges'].astype(np.int32), dev)
    l = imgops.sdf_reg_loss(sdf[:, None], e)
    assert abs(l.item() - float(g['reg'])) < 2e-6
    l.backward()
    assert _rel(sdf.grad, g['reg_dsdf']) < 1e-4


# ---- grid encoding + texture MLP ----------------------------------------------------------------------------------
def check_texmlp(dev, n=700):
    from d3h import texmlp
    from oracle import texmlp as OT
    gen = torch.Generator().manual_seed(31)
    npar = texmlp.grid_param_count()
    _, total = OT.grid_layout()
    assert npar == 2 * total
    table = (torch.rand(npar, generator=gen) * 2 - 1) * 0.5
    w1 = torch.randn(32, 10, generator=gen) * 0.5
    w2 = torch.randn(32, 32, generator=gen) * 0.3
    w3 = torch.randn(6, 32, generator=gen) * 0.3
    bbox = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)                                     # mlptexture.py:94 (sign-flipped box, kept literally)
    omin, omax = (0, 0, 0, 0, 0.001, 0), (1, 1, 1, 0, 1, 1)
    x = torch.rand(n, 3, generator=gen) * torch.tensor([1.8, 2.2, 0.6]) + torch.tensor([-1.0, -1.4, -0.3])   # partly outside the box
    x[0] = torch.tensor([-0.8, -1.2, -0.2])                                     # exactly on the x_n == 1 corner (level-0 wrap)
    mask = (torch.rand(n, generator=gen) > 0.2).float()
    if n >= 400:
        mask[128:352] = 0        # three background waves in a row, then a wave whose first 32-pixel MFMA chain is all background
    args = [t.clone().to(dev).requires_grad_(True) for t in (x, table, w1, w2, w3)]
    out = texmlp.texture_mlp(args[0], args[1], args[2], args[3], args[4], bbox, omin, omax, mask=mask.to(dev))
    ref_args = [t.clone().requires_grad_(True) for t in (x, table, w1, w2, w3)]
    ref = OT.texture_mlp(ref_args[0], ref_args[1], ref_args[2], ref_args[3], ref_args[4], bbox, omin, omax) * mask[:, None]
    assert (out.detach().cpu() - ref.detach()).abs().max() < 1e-5
    G = torch.randn(ref.shape, generator=gen)
    if n >= 200:
        G[40:136] = 0.0          # covered pixels WITHOUT an upstream gradient: a whole 32-pixel MFMA chain and a whole wave of them (their d(encoding) must be written as zero)
    (out * G.to(dev)).sum().backward()
    (ref * G).sum().backward()
    for a, r, name in zip(args, ref_args, ('x', 'table', 'w1', 'w2', 'w3')):
        d = (a.grad.cpu() - r.grad).abs().max() / (r.grad.abs().max() + 1e-12)
        assert d < 2e-4, (name, d.item())
    # the backward runs on the fp16 matrix pipe with a per-chain power-of-two gradient scale (csrc/texmlp.hip: texmlp_bwd_mlp_h2_kernel): an
    # upstream gradient of the magnitude a training step hands it (1e-7: a mean over 4 10^6 pixels) must give the same gradients, scaled
    args2 = [t.clone().to(dev).requires_grad_(True) for t in (x, table, w1, w2, w3)]
    out2 = texmlp.texture_mlp(args2[0], args2[1], args2[2], args2[3], args2[4], bbox, omin, omax, mask=mask.to(dev))
    (out2 * (G * 3e-7).to(dev)).sum().backward()
    for a, a2, name in zip(args, args2, ('x', 'table', 'w1', 'w2', 'w3')):
        d = (a2.grad.cpu() / 3e-7 - a.grad.cpu()).abs().max() / (a.grad.abs().max().cpu() + 1e-30)
        assert d < 2e-5, (name, 'tiny upstream gradient', d.item())
    # stand-alone encoding (tinycudann.Encoding.forward)
    xe = torch.rand(300, 3, generator=gen)
    xa, ta = xe.clone().to(dev).requires_grad_(True), table.clone().to(dev).requires_grad_(True)
    e = texmlp.grid_encode(xa, ta)
    xr, tr = xe.clone().requires_grad_(True), table.clone().requires_grad_(True)
    er = OT.grid_encode(xr, tr)
    assert (e.detach().cpu() - er.detach()).abs().max() < 1e-5     # fmaf(x, scale, 0.5) vs mul+add: 1 ulp of p ~ 4e-6 at scale 69
    G2 = torch.randn(er.shape, generator=gen)
    (e * G2.to(dev)).sum().backward()
    (er * G2).sum().backward()
    assert (ta.grad.cpu() - tr.grad).abs().max() < 5e-5 * max(1.0, tr.grad.abs().max().item())
    assert (xa.grad.cpu() - xr.grad).abs().max() < 5e-3 * xr.grad.abs().max()
    # spatially COHERENT points (what a rendered pixel row is): consecutive points sit in the same or a face-adjacent grid cell, the case the
    # table scatter reduces in the wave (runs of equal cells) and folds across faces (csrc/texmlp.hip: TexParams::merge_faces) before its atomics --
    # a curve that moves by ~1/8 of a finest-level cell per point, along each axis in turn and back again, diagonal stretches included
    m = 1024
    t_ = torch.arange(m, dtype=torch.float32) / m
    xc_ = torch.stack([0.1 + 0.8 * t_, 0.5 + 0.3 * torch.sin(9.0 * t_), 0.5 + 0.25 * torch.cos(5.0 * t_)], -1)
    xc_[m // 2:] = xc_[m // 2:].flip(0)[:, [1, 2, 0]]                       # second half: back along permuted axes
    xa, ta = xc_.clone().to(dev).requires_grad_(True), table.clone().to(dev).requires_grad_(True)
    e = texmlp.grid_encode(xa, ta)
    xr, tr = xc_.clone().requires_grad_(True), table.clone().requires_grad_(True)
    er = OT.grid_encode(xr, tr)
    G3 = torch.randn(er.shape, generator=gen)
    (e * G3.to(dev)).sum().backward()
    (er * G3).sum().backward()
    assert (ta.grad.cpu() - tr.grad).abs().max() < 5e-5 * max(1.0, tr.grad.abs().max().item()), float((ta.grad.cpu() - tr.grad).abs().max())
    assert float(tr.grad.abs().max()) > 3.0                                  # (many points per cell: the sums are not single contributions)


def check_texmlp_shared_table(dev, n=6000, passes=3, vs_oracle=True):
    """Several texture-MLP nodes on ONE table inside one backward pass (shade() samples twice for kd_grad / ks_grad, the split stage
    renders twice): the accumulated `.grad` of the table must be the sum of the contributions -- on a fresh leaf, with a pre-existing
    .grad, with a stand-alone encoding node on the same table, over repeated passes.  On the GPU the first contribution's scatter runs
    on a side stream and every later one joins it first (d3h/texmlp.py); the reference value there is the same computation with the
    side stream switched off (same kernels, so large n can be compared to atomic-order noise), at small n also the oracle.
    (Against the oracle a handful of points per 10^5 differ by O(1): a hidden unit whose pre-activation is within rounding of zero
    takes the other ReLU branch, so large-n comparisons with it are not meaningful point by point.)"""
    from d3h import texmlp
    from oracle import texmlp as OT
    gen = torch.Generator().manual_seed(77)
    npar = texmlp.grid_param_count()
    table = (torch.rand(npar, generator=gen) * 2 - 1) * 0.5
    w = [torch.randn(32, 10, generator=gen) * 0.5, torch.randn(32, 32, generator=gen) * 0.3, torch.randn(6, 32, generator=gen) * 0.3]
    bbox = (0.6, 0.6, 0.2, -0.8, -1.2, -0.2)
    omin, omax = (0, 0, 0, 0, 0.001, 0), (1, 1, 1, 0, 1, 1)
    xs = [torch.rand(n, 3, generator=gen) * torch.tensor([1.4, 1.8, 0.4]) + torch.tensor([-0.8, -1.2, -0.2]) for _ in range(3)]
    Gs = [torch.randn(n, 6, generator=gen) for _ in range(3)]
    enc_x = torch.rand(500, 3, generator=gen)

    def run(p_, tab, ws):
        xa = [x.clone().to(dev).requires_grad_(True) for x in xs]
        tab.grad = None if p_ % 3 != 1 else torch.full_like(tab, 0.25)       # every third pass: the leaf already holds a gradient
        for t in ws:
            t.grad = None
        total = sum((texmlp.texture_mlp(x, tab, ws[0], ws[1], ws[2], bbox, omin, omax) * G.to(dev)).sum() for x, G in zip(xa, Gs))
        if p_ % 3 == 2:                                                        # a stand-alone encoding node on the same table as well
            total = total + 0.0 * texmlp.grid_encode(enc_x.to(dev), tab).sum()
        total.backward()
        return tab.grad.detach().clone() - (0.25 if p_ % 3 == 1 else 0.0), [a.grad.detach().clone() for a in xa], [t.grad.detach().clone() for t in ws]
    tab = table.clone().to(dev).requires_grad_(True)
    ws = [t.clone().to(dev).requires_grad_(True) for t in w]
    was = texmlp.ASYNC_TABLE_GRAD
    try:
        texmlp.ASYNC_TABLE_GRAD = False
        ref = run(0, tab, ws)
        if dev != 'cpu':
            torch.cuda.synchronize()
        texmlp.ASYNC_TABLE_GRAD = True
        for p_ in range(passes):
            got = run(p_, tab, ws)
            den = ref[0].abs().max()
            assert (got[0] - ref[0]).abs().max() < 2e-5 * den, (p_, float((got[0] - ref[0]).abs().max() / den))
            for a, r in zip(got[1] + got[2], ref[1] + ref[2]):
                assert (a - r).abs().max() < 2e-5 * r.abs().max()
    finally:
        texmlp.ASYNC_TABLE_GRAD = was
    if vs_oracle:
        rt = table.clone().requires_grad_(True)
        rw = [t.clone().requires_grad_(True) for t in w]
        rx = [x.clone().requires_grad_(True) for x in xs]
        sum((OT.texture_mlp(x, rt, rw[0], rw[1], rw[2], bbox, omin, omax) * G).sum() for x, G in zip(rx, Gs)).backward()
        assert (ref[0].cpu() - rt.grad).abs().max() < 3e-4 * rt.grad.abs().max()
        for a, r in zip(ref[1] + ref[2], rx + rw):
            assert (a.cpu() - r.grad).abs().max() < 3e-4 * r.grad.abs().max()


# ---- render_mesh: the build's render.py against the REFERENCE's render.py (driven by the oracle dr / tcnn) -------------------------
def check_render_mesh_golden(dev):
    """all 12 buffers of render.render_mesh against the reference's render.py output, on any device (the random jitter is pre-drawn)"""
    from render import mesh as M, render as R
    from render.mlptexture import MLPTexture3D
    from oracle import texmlp as OT
    g = golden('render.npz')
    v, f = T(g['v'], dev), T(g['f'], dev)
    mn, mx = T(g['omin'], dev), T(g['omax'], dev)
    tex = MLPTexture3D((v.min(0).values, v.max(0).values), channels=6, min_max=[mn, mx]).to(dev)
    gen = torch.Generator().manual_seed(int(g['enc_seed']))
    with torch.no_grad():
        tex.encoder.params.copy_(((torch.rand(2 * OT.grid_layout()[1], generator=gen) * 2 - 1) * float(g['enc_scale'])).to(dev))
        for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
            tex.net.net[i].weight.copy_(T(g[k], dev))
    mat = {'kd_ks': tex, 'bsdf': 'pbr'}
    m = M.auto_normals(M.Mesh(v, f, material=mat))
    m_orig = M.auto_normals(M.Mesh(v * 0.97 + 0.01, f, material=mat))
    # the reference consumed the global CPU generator (seed 5) in this order: tangent noise, pixel offset, position noise
    # (render.py:285, :68, :84); the same draws are handed to the build on whatever device it runs
    from oracle import render as ORD
    torch.manual_seed(5)
    draws = ORD.draw_jitter(2, 48, 48)
    out = R.render_mesh(None, 0, None, m, m_orig, T(g['mvp'], dev), T(g['campos'], dev), None, [48, 48], spp=1, msaa=True,
                        background=T(g['bg'], dev), use_uv=False, extra_dict={'msdf': T(g['msdf'], dev)}, _rng_draws=draws)
    assert np.array_equal(out['visible_triangles'].cpu().numpy(), g['out.visible_triangles'])
    tol = {'z_grad': 2e-3, 'depth': 2e-4, 'invdepth': 2e-5}
    for k in ('shaded', 'z_grad', 'normal', 'geometric_normal', 'kd', 'ks', 'kd_grad', 'ks_grad', 'normal_grad', 'depth', 'invdepth', 'msdf_image'):
        a, b = out[k].detach().cpu().numpy(), g['out.' + k]
        assert a.shape == b.shape, k
        assert np.abs(a - b).max() < tol.get(k, 5e-5), (k, np.abs(a - b).max())
    # the supersampled path (render.py:239-245,334-336,424-451; spp = 2, msaa): visibility at 48^2, shading at 24^2, average-pooled output
    torch.manual_seed(6)
    draws = ORD.draw_jitter(2, 24, 24)
    out = R.render_mesh(None, 0, None, m, m_orig, T(g['mvp'], dev), T(g['campos'], dev), None, [24, 24], spp=2, msaa=True,
                        background=T(g['bg_spp2'], dev), use_uv=False, extra_dict={'msdf': T(g['msdf'], dev)}, _rng_draws=draws)
    assert np.array_equal(out['visible_triangles'].cpu().numpy(), g['out_spp2.visible_triangles'])
    for k in ('shaded', 'z_grad', 'normal', 'geometric_normal', 'kd', 'ks', 'kd_grad', 'ks_grad', 'normal_grad', 'depth', 'invdepth', 'msdf_image'):
        a, b = out[k].detach().cpu().numpy(), g['out_spp2.' + k]
        assert a.shape == b.shape, k
        assert np.abs(a - b).max() < tol.get(k, 5e-5), ('spp2', k, np.abs(a - b).max())
    # and it is differentiable end to end (the nearest resampling of the raster passes the gradient of the shaded pixels on)
    vv = v.clone().requires_grad_(True)
    m2 = M.auto_normals(M.Mesh(vv, f, material=mat))
    o2 = R.render_mesh(None, 0, None, m2, m_orig, T(g['mvp'], dev), T(g['campos'], dev), None, [24, 24], spp=2, msaa=True,
                       background=T(g['bg_spp2'], dev), use_uv=False, extra_dict={'msdf': T(g['msdf'], dev)}, _rng_draws=draws,
                       buffers=('shaded', 'geometric_normal'))
    (o2['shaded'].sum() + o2['geometric_normal'].square().sum()).backward()
    assert torch.isfinite(vv.grad).all() and vv.grad.abs().max() > 0


def check_render_uv(dev, res=40):
    """render.render_uv (render.py:456-472): the texture bake in uv space against the oracle composition rasterize -> interpolate -> MLP"""
    from render import mesh as M, render as R
    from render.mlptexture import MLPTexture3D
    from oracle import raster as OR, texmlp as OT
    g = golden('render.npz')
    v, f = T(g['v'], dev), T(g['f'], dev)
    lo, hi = v.min(0).values, v.max(0).values
    uv = ((v[:, :2] - lo[:2]) / (hi[:2] - lo[:2]) * 0.9 + 0.05).contiguous()
    mn, mx = T(g['omin'], dev), T(g['omax'], dev)
    tex = MLPTexture3D((lo, hi), channels=6, min_max=[mn, mx]).to(dev)
    gen = torch.Generator().manual_seed(int(g['enc_seed']))
    table = (torch.rand(2 * OT.grid_layout()[1], generator=gen) * 2 - 1) * float(g['enc_scale'])
    with torch.no_grad():
        tex.encoder.params.copy_(table.to(dev))
        for i, k in zip((0, 2, 4), ('w1', 'w2', 'w3')):
            tex.net.net[i].weight.copy_(T(g[k], dev))
    m = M.Mesh(v, f, v_tex=uv, t_tex_idx=f, material={'kd_ks': tex, 'bsdf': 'pbr'})
    cover, kd, ks = R.render_uv(None, m, [res, res], tex)
    vo, fo, uvo = v.cpu(), f.cpu().long(), uv.cpu()
    clip = torch.cat((uvo * 2 - 1, torch.zeros(vo.shape[0], 1), torch.ones(vo.shape[0], 1)), -1)[None]
    rast, _ = OR.rasterize(clip, fo, res, res)
    gp, _ = OR.interpolate(vo[None], rast, fo)
    ref = OT.texture_mlp(gp, table, T(g['w1'], 'cpu'), T(g['w2'], 'cpu'), T(g['w3'], 'cpu'), (0.6, 0.6, 0.2, -0.8, -1.2, -0.2), g['omin'].tolist(), g['omax'].tolist())
    cov_o = (rast[..., 3:] > 0).float()
    assert float(cov_o.mean()) > 0.1
    same = (cover.cpu() == cov_o).float().mean()
    assert same > 0.999, float(same)                      # texels exactly on a uv-chart fold may go either way
    both = (cover.cpu() * cov_o) > 0
    assert ((kd.detach().cpu() - ref[..., 0:3]).abs() * both).max() < 2e-4
    assert ((ks.detach().cpu() - ref[..., 3:6]).abs() * both).max() < 2e-4


def check_fused_adam(dev, steps=6):
    """d3h.optim.FusedAdam (one launch for all tensors, gradient scale and clamp folded in) == torch.optim.Adam + the reference's
    explicit `grad /= 8` and clamp kernels around it, over several steps with a changing learning rate and a tensor that gets no
    gradient in some steps"""
    from d3h import optim as O
    gen = torch.Generator().manual_seed(3)
    shapes = [(256, 39), (256,), (5, 256, 256), (1, 256), (3001, 3), (777,), (40000, 2), (1,)] + [(17,)] * 30      # > 32 tensors: two launches
    pa = [torch.randn(s, generator=gen).to(dev).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    ga = [{'params': pa[:4], 'lr': 3e-4}, {'params': pa[4:6], 'lr': 0.03}, {'params': pa[6:], 'lr': 0.005}]
    gb = [{'params': pb[:4], 'lr': 3e-4}, {'params': pb[4:6], 'lr': 0.03}, {'params': pb[6:], 'lr': 0.005}]
    fa = O.FusedAdam(ga)
    fa.set_grad_scale(pa[6], 1.0 / 8.0)
    fa.set_clamp(pa[4], -1.0, 1.0)
    tb = torch.optim.Adam(gb, eps=1e-8)
    sched = O.lr_schedule(3)
    sa, sb = O.LambdaLR(fa, sched), torch.optim.lr_scheduler.LambdaLR(tb, lr_lambda=lambda k: sched(k))
    for it in range(steps):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 5 and it % 2 == 1:
                a.grad = b.grad = None                      # no gradient this step: both skip the tensor (its step count does not advance)
                continue
            gr = torch.randn(a.shape, generator=gen) * (10.0 if i == 4 else 1.0)
            a.grad, b.grad = gr.to(dev), gr.to(dev).clone()
        pb[6].grad /= 8.0
        ver = [p._version for p in pa]
        fa.step(); sa.step()
        tb.step(); sb.step()
        # the raw-pointer update must be visible to autograd / `_version`-keyed caches exactly as torch's in-place step is
        assert all((p._version > v0) == (p.grad is not None) for p, v0 in zip(pa, ver)), 'FusedAdam.step() must bump _version of what it stepped'
        with torch.no_grad():
            pb[4].clamp_(-1.0, 1.0)
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert (a.detach() - b.detach()).abs().max() <= 2e-6 * max(1.0, float(b.detach().abs().max())), (it, i, float((a - b).abs().max()))
    assert float(pa[4].detach().abs().max()) <= 1.0
    # a NaN gradient must surface as NaN in the clamped tensor, as torch.clamp after torch's Adam gives (a fminf / fmaxf clamp hides it)
    for a, b in zip(pa, pb):
        a.grad, b.grad = torch.zeros_like(a), torch.zeros_like(b)
    pa[4].grad[7, 1] = pb[4].grad[7, 1] = float('nan')
    fa.step(); tb.step()
    with torch.no_grad():
        pb[4].clamp_(-1.0, 1.0)
    assert torch.isnan(pa[4][7, 1]) and torch.isnan(pb[4][7, 1])
    assert torch.equal(torch.isnan(pa[4].detach()), torch.isnan(pb[4].detach()))
    # a gradient of the wrong dtype (a foreign producer / hook) must be refused, not reinterpreted as float* 
    import pytest
    for a in pa:
        a.grad = torch.zeros_like(a)
    if hasattr(pa[1], 'grad_dtype'):
        pa[1].grad_dtype = None                              # (torch >= 2.9 refuses the assignment itself unless told otherwise)
    pa[1].grad = torch.zeros(pa[1].shape, dtype=torch.float64, device=pa[1].device)
    before = pa[0].detach().clone()
    with pytest.raises(RuntimeError, match='FusedAdam: gradient'):
        fa.step()
    assert torch.equal(pa[0].detach(), before)               # nothing was launched


def check_smplx_pose_kernel(dev, nb=5):
    """csrc/smplx_pose.hip (Rodrigues + kinematic chain + rest-pose removal, forward and backward) against the oracle
    (oracle/lbs.py <- deform/smplx_exavatar/lbs.py:311-413): A, d(full pose), d(rest joints); also through SMPLX.transforms on the
    miniature model of the LBS golden (A of the reference itself)"""
    from d3h import smplx_pose as SP, synth
    import oracle.lbs as OL
    gen = torch.Generator().manual_seed(9)
    par = torch.tensor(synth.PARENTS, dtype=torch.long)
    fp = torch.randn(nb, 55, 3, generator=gen) * 0.4
    fp[:, 23:] = 0                                                      # body_models.py:1255: entries >= 69 of the full pose are zero
    fp[0, 3] = 0                                                        # an exactly-zero rotation inside the used range
    J = torch.from_numpy(synth.rest_joints())[None] + 0.01 * torch.randn(nb, 55, 3, generator=gen)
    fo, Jo = fp.clone().requires_grad_(True), J.clone().requires_grad_(True)
    rot = OL.batch_rodrigues(fo.view(-1, 3)).view(nb, 55, 3, 3)
    _, Ao = OL.rigid_chain(rot, Jo, par)
    fa, Ja = fp.clone().to(dev).requires_grad_(True), J.clone().to(dev).requires_grad_(True)
    p32 = torch.tensor([max(p, 0) for p in synth.PARENTS], dtype=torch.int32, device=dev)
    A = SP.pose_transforms(fa, Ja, p32)
    assert (A.detach().cpu() - Ao.detach()).abs().max() < 2e-6
    W = torch.randn(nb, 55, 4, 4, generator=gen)
    (A * W.to(dev)).sum().backward()
    (Ao * W).sum().backward()
    assert (fa.grad.cpu() - fo.grad)[:, :23].abs().max() < 2e-5 * fo.grad.abs().max(), float((fa.grad.cpu() - fo.grad)[:, :23].abs().max())
    assert (Ja.grad.cpu() - Jo.grad).abs().max() < 2e-5 * Jo.grad.abs().max()
    # shared rest joints (one row for all frames): the gradient is summed over the frames
    J1 = J[:1].clone().to(dev).requires_grad_(True)
    A1 = SP.pose_transforms(fp.to(dev), J1, p32)
    (A1 * W.to(dev)).sum().backward()
    J1o = J[:1].clone().requires_grad_(True)
    _, A1o = OL.rigid_chain(OL.batch_rodrigues(fp.view(-1, 3)).view(nb, 55, 3, 3), J1o.expand(nb, -1, -1), par)
    (A1o * W).sum().backward()
    assert (J1.grad.cpu() - J1o.grad).abs().max() < 2e-5 * J1o.grad.abs().max()
    # through the layer: kernel path == level-batched torch path == reference golden
    g, d = _lbs_setup(dev)
    betas = T(g['betas'], dev)
    args = (betas, T(g['root_pose'], dev), T(g['body_pose'], dev), T(g['jaw'], dev), T(g['expr'], dev), T(g['face_offset'], dev), T(g['joint_offset'], dev),
            T(g['locator_offset'], dev))
    Ak, Ar = d.layer.transforms(*args), d.layer.transforms_reference(*args)
    assert (Ak - Ar).abs().max() < 2e-6 and (Ak.cpu() - torch.from_numpy(g['A'])).abs().max() < 2e-6


def check_rasterize_near_plane(dev, res=48):
    """triangles crossing the camera plane (w <= 0 at one or two vertices): covered pixels, barycentrics, z/w and the position gradient
    against the oracle's homogeneous rasterisation; interpolation of a crossing triangle; antialias leaves such triangles alone"""
    from d3h import raster
    from oracle import raster as OR
    n, f = 0.1, 10.0
    P = np.array([[1.2, 0, 0, 0], [0, 1.2, 0, 0], [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)], [0, 0, -1, 0]], np.float32)
    v = np.array([[-0.5, -0.4, -2.0], [0.5, -0.4, -2.0], [0.6, -0.2, 1.0], [-0.6, -0.2, 1.0], [0, 0.3, -1.5], [0.3, 0.5, -1.5], [-0.3, 0.5, -1.5],
                  [0.2, 0.1, -3.0], [0.9, 0.1, 0.5], [0.9, 0.6, -3.0]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [7, 8, 9]], np.int64)          # two crossing (one vertex behind / two behind), one ordinary, one more crossing
    vh = (np.concatenate([v, np.ones((len(v), 1), np.float32)], 1) @ P.T)[None].astype(np.float32)
    assert (vh[0, :, 3] <= 0).any()
    pos, pos_o = T(vh, dev, True), torch.from_numpy(vh).requires_grad_(True)
    rast, db = raster.rasterize(pos, T(tri.astype(np.int32), dev), (res, res))
    rast_o, db_o = OR.rasterize(pos_o, torch.from_numpy(tri), res, res)
    r, ro = rast.detach().cpu(), rast_o.detach()
    assert torch.equal(r[..., 3], ro[..., 3]), f'{(r[..., 3] != ro[..., 3]).sum().item()} pixels differ in triangle id'
    for t in (1, 2, 3, 4):
        assert int((ro[..., 3] == t).sum()) > 10, t                              # every triangle is visible, the crossing ones too
    assert torch.isfinite(r).all() and (r[..., :3] - ro[..., :3]).abs().max() < 2e-4
    gen = torch.Generator().manual_seed(5)
    G = torch.randn(r.shape, generator=gen)
    G[..., 2:] = 0
    (rast * G.to(dev)).sum().backward()
    (rast_o * G).sum().backward()
    assert torch.isfinite(pos.grad).all()
    assert (pos.grad.cpu() - pos_o.grad).abs().max() < 2e-3 * pos_o.grad.abs().max()
    col = torch.rand(1, res, res, 3, generator=gen)
    aa = raster.antialias(col.to(dev), rast_o.to(dev), T(vh, dev), T(tri.astype(np.int32), dev))
    aa_o = OR.antialias(col, rast_o, torch.from_numpy(vh), torch.from_numpy(tri))
    assert (aa.cpu() - aa_o).abs().max() < 1e-5
