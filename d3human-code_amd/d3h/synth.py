"""Seeded synthetic stand-ins for the licence-gated / un-shipped inputs of the reference:
an "SMPL-X-shaped" body model (V=10475, J=55, real kinematic tree), Kuhn tet grids, poses, camera.
The real SMPL-X files load through deform.smplx_exavatar.body_models when present; these generators are what
bench.py, __graft_entry__.smoke() and the tests use (no network, no datasets in this environment).
"""
import itertools
import math

import numpy as np
import torch

# SMPL-X kinematic parents (55 joints; body 22, jaw, eyes, 15+15 hand joints) -- deform/smplx_exavatar/body_models.py:264-266
PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
           20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
           21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def rest_joints():
    J = np.zeros((55, 3), np.float32)
    body = {0: (0, -0.30, 0), 1: (0.09, -0.38, 0), 2: (-0.09, -0.38, 0), 3: (0, -0.18, 0), 4: (0.10, -0.78, 0),
            5: (-0.10, -0.78, 0), 6: (0, -0.05, 0), 7: (0.10, -1.18, 0), 8: (-0.10, -1.18, 0), 9: (0, 0.07, 0),
            10: (0.10, -1.23, 0.12), 11: (-0.10, -1.23, 0.12), 12: (0, 0.24, 0), 13: (0.07, 0.16, 0), 14: (-0.07, 0.16, 0),
            15: (0, 0.34, 0), 16: (0.19, 0.18, 0), 17: (-0.19, 0.18, 0), 18: (0.45, 0.18, 0), 19: (-0.45, 0.18, 0),
            20: (0.70, 0.18, 0), 21: (-0.70, 0.18, 0), 22: (0, 0.31, 0.04), 23: (0.03, 0.39, 0.08), 24: (-0.03, 0.39, 0.08)}
    for k, v in body.items():
        J[k] = v
    for side, base, sgn in ((0, 25, 1.0), (1, 40, -1.0)):
        for f in range(5):
            for k in range(3):
                J[base + 3 * f + k] = (sgn * (0.75 + 0.03 * k + 0.005 * f), 0.18, 0.04 - 0.02 * f)
    return J


def make_body_model(n_verts=10475, seed=0, n_shape=100, n_expr=50):
    """dict with the fields of an SMPL-X .npz that the hot path consumes (v_template, weights, J_regressor, shapedirs,
    expr_dirs, posedirs, parents) -- seeded, smooth, sparse (top-4) skin weights."""
    rng = np.random.default_rng(seed)
    J = rest_joints()
    par = np.array(PARENTS)
    # bones: child joint -> parent joint; vertices sampled on capsule surfaces around the bones
    bones = [(j, par[j]) for j in range(1, 55)]
    rad = np.full(55, 0.012, np.float32)
    rad[:22] = [0.12, 0.075, 0.075, 0.13, 0.055, 0.055, 0.14, 0.045, 0.045, 0.14, 0.04, 0.04, 0.06, 0.07, 0.07, 0.10,
                0.055, 0.055, 0.045, 0.045, 0.035, 0.035]
    lens = np.array([np.linalg.norm(J[c] - J[p]) + 0.02 for c, p in bones])
    wgt = lens * np.array([rad[c] for c, _ in bones])
    cnt = np.maximum(2, np.floor(wgt / wgt.sum() * n_verts * 0.9)).astype(int)
    cnt[np.argmax(cnt)] += n_verts - cnt.sum()
    verts = []
    for (c, p), m in zip(bones, cnt):
        a, b = J[p], J[c]
        t = rng.random(m)[:, None]
        d = rng.normal(size=(m, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        verts.append(a + t * (b - a) + d * rad[c])
    v = np.concatenate(verts, 0).astype(np.float32)[:n_verts]
    # skin weights: softmax(-d^2/sigma^2) to joints, top-4
    d2 = ((v[:, None, :] - J[None]) ** 2).sum(-1)
    w = np.exp(-(d2 - d2.min(1, keepdims=True)) / (0.08 ** 2))
    th = np.sort(w, axis=1)[:, -4][:, None]
    w = np.where(w >= th, w, 0.0)
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    # joint regressor: weight-proportional vertex average, shifted so that J_regressor @ v_template == rest joints approx
    Jr = (w / np.maximum(w.sum(0, keepdims=True), 1e-8)).T.astype(np.float32)          # [55, V]
    shapedirs = (rng.normal(size=(n_verts, 3, n_shape)) * 1e-3).astype(np.float32)
    expr_dirs = (rng.normal(size=(n_verts, 3, n_expr)) * 1e-3).astype(np.float32)
    posedirs = (rng.normal(size=(54 * 9, n_verts * 3)) * 1e-3).astype(np.float32)
    return {'v_template': v, 'weights': w, 'J_regressor': Jr, 'shapedirs': shapedirs, 'expr_dirs': expr_dirs,
            'posedirs': posedirs, 'parents': par.astype(np.int64)}


def kuhn_grid(n, raw=False):
    """(n+1)^3 vertices, 6 n^3 tets (Kuhn subdivision), then the reference's y -= 0.1919; *= 1.2 (geometry/hmsdf.py:210-211).
    Every tet is positively oriented (((b-a) x (c-a)) . (d-a) > 0), as tetgen / quartet grids are: the marching-tets triangle table
    (gshell_tets.py:91-203) then yields consistently wound, outward-facing triangles."""
    g = np.arange(n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32) / n * 2 - 1
    if not raw:
        verts[:, 1] -= np.float32(0.1919)
        verts *= np.float32(1.2)
    vid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k
    c = np.arange(n)
    I, J, K = [a.reshape(-1) for a in np.meshgrid(c, c, c, indexing='ij')]
    tets = []
    for perm in itertools.permutations(range(3)):
        cur = [I.copy(), J.copy(), K.copy()]
        ids = [vid(*cur)]
        for ax in perm:
            cur[ax] = cur[ax] + 1
            ids.append(vid(*cur))
        t = np.stack(ids, -1)
        if np.linalg.det(np.eye(3)[list(perm)]) < 0:        # odd permutations walk the cube with the opposite handedness
            t = t[:, [0, 1, 3, 2]]
        tets.append(t)
    return verts, np.stack(tets, 1).reshape(-1, 4).astype(np.int64)


def body_sdf(x, joints=None, scale=1.0):
    """analytic SDF (positive outside) of the capsule humanoid; x: [N,3] torch tensor"""
    J = torch.as_tensor(rest_joints() if joints is None else joints, dtype=x.dtype, device=x.device)
    rad = torch.full((55,), 0.012, dtype=x.dtype, device=x.device)
    rad[:22] = torch.tensor([0.12, 0.075, 0.075, 0.13, 0.055, 0.055, 0.14, 0.045, 0.045, 0.14, 0.04, 0.04, 0.06, 0.07, 0.07,
                             0.10, 0.055, 0.055, 0.045, 0.045, 0.035, 0.035], dtype=x.dtype, device=x.device)
    out = torch.full((x.shape[0],), 1e9, dtype=x.dtype, device=x.device)
    for c in range(1, 25):            # body + head joints (fingers are below grid resolution)
        a, b = J[PARENTS[c]], J[c]
        ab = b - a
        t = ((x - a) @ ab / (ab @ ab + 1e-12)).clamp(0, 1)
        d = (x - (a + t[:, None] * ab)).norm(dim=-1) - rad[c]
        out = torch.minimum(out, d)
    return out * scale


def camera(res, fx_scale=1.2, dist=2.5, n=0.001, f=1000.0):
    """projection of dataset/dataset_split.py:57-68 (get_ndc_matrix_from_ss) and mv = diag(1,-1,-1,1) w2c (:181-194)."""
    H = W = res
    fx = fy = fx_scale * W
    cx = cy = W / 2
    proj = np.zeros((4, 4), np.float32)
    proj[0, 0] = 2 * fx / (W - 1); proj[0, 2] = 1 - 2 * cx / (W - 1)
    proj[1, 1] = -2 * fy / (H - 1); proj[1, 2] = 1 - 2 * cy / (H - 1)
    proj[2, 2] = -(f + n) / (f - n); proj[2, 3] = -2 * f * n / (f - n)
    proj[3, 2] = -1
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, 3] = (0, -0.43, dist)         # body centre (y ~ -0.43) on the optical axis, dist in front (OpenCV: +z forward, +y down)
    w2c[1, 1] = -1                        # world is y-up, OpenCV camera is y-down
    w2c[2, 2] = -1
    mv = np.diag([1, -1, -1, 1]).astype(np.float32) @ w2c
    mvp = proj @ mv
    campos = np.linalg.inv(mv)[:3, 3]
    return mv, mvp, campos.astype(np.float32)


def poses(n_frames, seed=1234):
    out = []
    for fidx in range(n_frames):
        g = torch.Generator().manual_seed(seed + fidx)
        out.append(torch.randn(63, generator=g) * 0.2)
    return torch.stack(out)


def write_tet_grid(path, n, positively_oriented=True):
    """Write the tet-grid file the reference loads at start-up (`data/tets/tet_grid.npz`, geometry/hmsdf.py:207: keys `vertices`
    float32 [N,3], `indices` int64 [T,4]) for a Kuhn lattice of n^3 cubes.  The reference's repository does not ship that file (it
    comes from tetgen, script/get_tet_smpl.py); coordinates are stored BEFORE the `y -= 0.1919; *= 1.2` of hmsdf.py:210-211, so a geometry
    built from the file has exactly the vertices of kuhn_grid(n)."""
    import os
    v, t = kuhn_grid(n, raw=True)          # the loader applies the offset and the scale: same float32 operations as kuhn_grid(n)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, vertices=v.astype(np.float32), indices=np.asarray(t, np.int64))
    return v.shape[0], t.shape[0]


def icosphere(sub):
    """closed manifold triangle mesh: subdivided octahedron projected on the unit sphere"""
    v = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    f = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    v = [np.array(p, np.float64) for p in v]
    for _ in range(sub):
        mid, nf = {}, []

        def m(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                p = v[a] + v[b]
                v.append(p / np.linalg.norm(p))
                mid[k] = len(v) - 1
            return mid[k]
        for a, b, c in f:
            ab, bc, ca = m(a, b), m(b, c), m(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, np.float32), np.array(f, np.int64)


def tube(n_theta, n_y):
    """open unit tube: x = cos t, z = sin t, y in [-1, 1]; n_theta x n_y vertices, 2 n_theta (n_y - 1) outward-wound faces"""
    t = np.arange(n_theta) * (2 * np.pi / n_theta)
    ys = np.linspace(-1.0, 1.0, n_y)
    v = np.array([(np.cos(a), y, np.sin(a)) for y in ys for a in t], np.float32)
    f = []
    for j in range(n_y - 1):
        for i in range(n_theta):
            a, b = j * n_theta + i, j * n_theta + (i + 1) % n_theta
            c, d = a + n_theta, b + n_theta
            f += [(a, c, b), (b, c, d)]
    return v, np.array(f, np.int64)


def refine_mesh_to(v, f, n_target):
    """edge splits (one new vertex, two new faces each; winding kept, the surface stays a closed manifold) until the mesh has exactly
    n_target vertices"""
    v = [np.asarray(p, np.float64) for p in v]
    f = [tuple(int(i) for i in t) for t in f]
    while len(v) < n_target:
        emap = {}
        for i, (a, b, c) in enumerate(f):
            for x, y in ((a, b), (b, c), (c, a)):
                emap.setdefault((min(x, y), max(x, y)), []).append(i)
        used, out = set(), {}
        for (x, y), fs in sorted(emap.items()):
            if len(v) >= n_target:
                break
            if len(fs) != 2 or fs[0] in used or fs[1] in used:
                continue
            used.update(fs)
            m = len(v)
            v.append(0.5 * (v[x] + v[y]))
            for fi in fs:
                a, b, c = f[fi]
                for p, q, r in ((a, b, c), (b, c, a), (c, a, b)):          # the rotation in which (p, q) is the split edge
                    if {p, q} == {x, y}:
                        out[fi] = [(p, m, r), (m, q, r)]
                        break
        nf = []
        for i, t in enumerate(f):
            nf += out.get(i, [t])
        f = nf
    return np.array(v, np.float32), np.array(f, np.int64)


def write_smplx_npz(path, n_verts=10475, seed=0):
    """Write a SYNTHETIC body model in the layout of the official, licence-gated `SMPLX_{GENDER}.npz` that
    deform/smplx_exavatar/body_models.py:976-1078 loads (np.load(..., allow_pickle=True)): `v_template` [V,3], `f` [F,3] uint32, `weights`
    [V,55], `J_regressor` [55,V], `shapedirs` [V,3,400] (300 shape + 100 expression components: SHAPE_SPACE_DIM / EXPRESSION_SPACE_DIM),
    `posedirs` [V,3,486], `kintree_table` [2,55] uint32 with 2^32 - 1 as the root's parent, and the hand-PCA / landmark keys the reference's
    constructor touches.  The surface is a closed ellipsoidal blob refined to exactly n_verts vertices around the SMPL-X joint layout of
    rest_joints() -- good for exercising the file path end to end (template mesh -> SDF pre-fit -> skinning), not an anatomical body.
    -> the dict of arrays written."""
    import os
    rng = np.random.default_rng(seed)
    s, f = icosphere(5)                                                       # 4 098 vertices / 8 192 faces
    s, f = refine_mesh_to(s, f, n_verts)
    s = s / np.linalg.norm(s, axis=1, keepdims=True)
    v = (s * np.array([0.33, 0.78, 0.24], np.float32) + np.array([0.0, -0.38, 0.0], np.float32)).astype(np.float32)
    J = rest_joints()
    d2 = ((v[:, None, :] - J[None]) ** 2).sum(-1)
    w = np.exp(-(d2 - d2.min(1, keepdims=True)) / (0.08 ** 2))
    th = np.sort(w, axis=1)[:, -4][:, None]
    w = np.where(w >= th, w, 0.0)
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    # J_regressor rows: sparse convex combinations of surface vertices whose centroid is near the joint
    Jr = np.zeros((55, n_verts), np.float32)
    for j in range(55):
        near = np.argsort(d2[:, j])[:32]
        Jr[j, near] = 1.0 / 32
    shapedirs = np.zeros((n_verts, 3, 400), np.float32)
    shapedirs[:, :, :100] = rng.normal(size=(n_verts, 3, 100)) * 1e-3
    shapedirs[:, :, 300:350] = rng.normal(size=(n_verts, 3, 50)) * 1e-3
    posedirs = (rng.normal(size=(n_verts, 3, 486)) * 1e-3).astype(np.float32)
    kin = np.stack([np.array(PARENTS, np.int64), np.arange(55)]).astype(np.int64)
    kin[0, 0] = 2 ** 32 - 1
    d = {'v_template': v, 'f': f.astype(np.uint32), 'weights': w, 'J_regressor': Jr, 'shapedirs': shapedirs, 'posedirs': posedirs,
         'kintree_table': kin.astype(np.uint32), 'hands_componentsl': np.eye(45, dtype=np.float32), 'hands_componentsr': np.eye(45, dtype=np.float32),
         'hands_meanl': np.zeros(45, np.float32), 'hands_meanr': np.zeros(45, np.float32),
         'lmk_faces_idx': np.arange(51, dtype=np.int64), 'lmk_bary_coords': np.full((51, 3), 1.0 / 3, np.float32),
         'dynamic_lmk_faces_idx': np.zeros((79, 17), np.int64), 'dynamic_lmk_bary_coords': np.full((79, 17, 3), 1.0 / 3, np.float32)}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.savez(path, **d)
    return d
