"""Oracle for G-Shell marching tetrahedra (torch CPU; differentiable through autograd).

Restates geometry/gshell_tets.py:253-447 (GShell_Tets.__call__) and geometry/hmsdf_tets_split.py:254-454
(hmSDF_Tets.__call__, identical except mSDF is negated for type == "body", :261-264), including
auto_normals (:9-33), compute_tangents (:40-83) and map_uv (:219-248).  The lookup tables are the data
constants of gshell_tets.py:91-203.  Pinned by tests/golden/mtets_*.npz (outputs of the reference
itself on seeded Kuhn grids).  TEST INFRASTRUCTURE -- never imported by the product.
"""
import math

import numpy as np
import torch

BASE_TET_EDGES = [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3]
NUM_TRIANGLES = [0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0]
TRIANGLE_TABLE = [
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, -1, -1, -1], [4, 0, 3, -1, -1, -1], [1, 4, 2, 1, 3, 4],
    [3, 1, 5, -1, -1, -1], [2, 3, 0, 2, 5, 3], [1, 4, 0, 1, 5, 4], [4, 2, 5, -1, -1, -1],
    [4, 5, 2, -1, -1, -1], [4, 1, 0, 4, 5, 1], [3, 2, 0, 3, 5, 2], [1, 3, 5, -1, -1, -1],
    [4, 1, 2, 4, 3, 1], [3, 0, 4, -1, -1, -1], [2, 0, 1, -1, -1, -1], [-1, -1, -1, -1, -1, -1]]
MESH_EDGE_TABLE = [
    [-1, -1, -1, -1, -1, -1], [1, 0, 2, 1, -1, -1], [4, 0, 3, 4, -1, -1], [1, 3, 4, 2, 1, -1],
    [3, 1, 5, 3, -1, -1], [2, 5, 3, 0, 2, -1], [1, 5, 4, 0, 1, -1], [4, 2, 5, 4, -1, -1],
    [4, 5, 2, 4, -1, -1], [4, 5, 1, 0, 4, -1], [3, 5, 2, 0, 3, -1], [1, 3, 5, 1, -1, -1],
    [4, 3, 1, 2, 4, -1], [3, 0, 4, 3, -1, -1], [2, 0, 1, 2, -1, -1], [-1, -1, -1, -1, -1, -1]]
NUM_TRIANGLES_TRI = [0, 1, 1, 2, 1, 2, 2, 1]
NUM_TRIANGLES_QUAD = [0, 1, 1, 2, 1, 4, 2, 3, 1, 2, 4, 3, 2, 3, 3, 2]
TRIANGLE_TABLE_TRI = [
    [-1, -1, -1, -1, -1, -1], [4, 2, 5, -1, -1, -1], [3, 1, 4, -1, -1, -1], [3, 1, 2, 3, 2, 5],
    [0, 3, 5, -1, -1, -1], [0, 3, 4, 0, 4, 2], [0, 1, 4, 0, 4, 5], [0, 1, 2, -1, -1, -1]]
_N = -1
TRIANGLE_TABLE_QUAD = [
    [_N] * 12,
    [6, 3, 7] + [_N] * 9,
    [5, 2, 6] + [_N] * 9,
    [5, 2, 7, 3, 7, 2] + [_N] * 6,
    [4, 1, 5] + [_N] * 9,
    [4, 1, 5, 4, 5, 7, 5, 6, 7, 7, 6, 3],
    [4, 1, 2, 6, 4, 2] + [_N] * 6,
    [4, 1, 2, 7, 4, 2, 7, 2, 3] + [_N] * 3,
    [0, 4, 7] + [_N] * 9,
    [0, 4, 6, 3, 0, 6] + [_N] * 6,
    [0, 4, 5, 0, 5, 2, 0, 2, 6, 0, 6, 7],
    [0, 4, 5, 0, 5, 2, 0, 2, 3] + [_N] * 3,
    [0, 1, 5, 7, 0, 5] + [_N] * 6,
    [0, 1, 5, 0, 5, 6, 0, 6, 3] + [_N] * 3,
    [0, 1, 2, 0, 2, 6, 0, 6, 7] + [_N] * 3,
    [0, 1, 2, 0, 2, 3] + [_N] * 6]


def _t(x):
    return torch.tensor(x, dtype=torch.long)


def _dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def _safe_normalize(x, eps=1e-20):
    # render/util.py:25-29: x / sqrt(clamp(dot(x,x), min=eps))
    return x / torch.sqrt(torch.clamp(_dot(x, x), min=eps))


def auto_normals(v_pos, faces):
    """gshell_tets.py:9-33 == render/mesh.py:418-446: area-weighted vertex normals."""
    v0, v1, v2 = v_pos[faces[:, 0]], v_pos[faces[:, 1]], v_pos[faces[:, 2]]
    fn = torch.cross(v1 - v0, v2 - v0, dim=-1)
    vn = torch.zeros_like(v_pos)
    for c in range(3):
        vn = vn.index_add(0, faces[:, c], fn)
    vn = torch.where(_dot(vn, vn) > 1e-20, vn, torch.tensor([0.0, 0.0, 1.0]))
    return _safe_normalize(vn)


def map_uv(face_gidx, max_idx):
    """gshell_tets.py:219-248."""
    n = int(np.ceil(np.sqrt((max_idx + 1) // 2)))
    lin = torch.linspace(0, 1 - (1 / n), n, dtype=torch.float32)
    ty, tx = torch.meshgrid(lin, lin, indexing='ij')
    pad = 0.9 / n
    uvs = torch.stack([tx, ty, tx + pad, ty, tx + pad, ty + pad, tx, ty + pad], dim=-1).view(-1, 2)
    tet = torch.div(face_gidx, 2, rounding_mode='trunc')
    tri = face_gidx % 2
    uv_idx = torch.stack((tet * 4, tet * 4 + tri + 1, tet * 4 + tri + 2), dim=-1).view(-1, 3)
    return uvs, uv_idx


def compute_tangents(v_pos, v_tex, v_nrm, t_pos, t_tex):
    """gshell_tets.py:40-83 (t_nrm_idx == t_pos_idx at the only call site, :327)."""
    pos = [v_pos[t_pos[:, i]] for i in range(3)]
    tex = [v_tex[t_tex[:, i]] for i in range(3)]
    uve1, uve2 = tex[1] - tex[0], tex[2] - tex[0]
    pe1, pe2 = pos[1] - pos[0], pos[2] - pos[0]
    nom = pe1 * uve2[..., 1:2] - pe2 * uve1[..., 1:2]
    den = uve1[..., 0:1] * uve2[..., 1:2] - uve1[..., 1:2] * uve2[..., 0:1]
    tang = nom / torch.where(den > 0.0, torch.clamp(den, min=1e-6), torch.clamp(den, max=-1e-6))
    tangents = torch.zeros_like(v_nrm)
    tansum = torch.zeros_like(v_nrm)
    for i in range(3):
        tangents = tangents.index_add(0, t_pos[:, i], tang)
        tansum = tansum.index_add(0, t_pos[:, i], torch.ones_like(tang))
    tangents = tangents / tansum
    tangents = _safe_normalize(tangents)
    return _safe_normalize(tangents - _dot(tangents, v_nrm) * v_nrm)


def _edge_weights(s_pair):
    """:291-300 -- s_pair [E,2] -> weights [E,2] = (-s1/d, s0/d), d = s0 - s1 regularised."""
    a = s_pair[:, 0]
    b = -s_pair[:, 1]
    den = a + b
    den = torch.sign(den) * (den.abs() + 1e-12)
    den = torch.where(den == 0, torch.full_like(den, 1e-12), den)
    return torch.stack([b / den, a / den], dim=1)


def gshell_tets(pos, sdf, msdf, tets, negate_msdf=False):
    """Returns a dict with every tensor the reference returns (+ a few intermediates used by tests)."""
    sdf = sdf.reshape(-1)
    if sdf.dtype != torch.float64:          # (a float64 evaluation keeps its values; its discrete decisions are taken on their float32 rounding)
        sdf = sdf.float()
    if negate_msdf:                       # hmsdf_tets_split.py:261-264 (type == "body"): negated INSIDE no_grad,
        msdf = (-msdf).detach()           # so the body pass sends no gradient to msdf (reference quirk, kept)
    with torch.no_grad():
        occ = sdf.float() > 0
        occ4 = occ[tets]
        nocc = occ4.sum(-1)
        valid = (nocc > 0) & (nocc < 4)                                   # :271-272 (watertight template)
        vt = tets[valid]
        e = vt[:, _t(BASE_TET_EDGES)].reshape(-1, 2)
        e = torch.stack([torch.minimum(e[:, 0], e[:, 1]), torch.maximum(e[:, 0], e[:, 1])], -1)   # sort_edges :209-217
        uniq, inv = torch.unique(e, dim=0, return_inverse=True)          # lexicographic row order == vertex ids
        cross = occ[uniq[:, 0]] != occ[uniq[:, 1]]
        vid = torch.full((uniq.shape[0],), -1, dtype=torch.long)
        vid[cross] = torch.arange(int(cross.sum()))
        idx_map = vid[inv].reshape(-1, 6)
        ev = uniq[cross]
        case = (occ4[valid].long() * _t([1, 2, 4, 8])).sum(-1)
        ntri = _t(NUM_TRIANGLES)[case]
        gid = torch.arange(tets.shape[0])[valid]

    w = _edge_weights(sdf[ev])                                           # [P_wt, 2]
    verts = pos[ev[:, 0]] * w[:, 0:1] + pos[ev[:, 1]] * w[:, 1:2]
    m_pair = msdf[ev]
    msdf_vert = m_pair[:, 0] * w[:, 0] + m_pair[:, 1] * w[:, 1]
    wd = w.detach()
    msdf_vert_sg = m_pair[:, 0] * wd[:, 0] + m_pair[:, 1] * wd[:, 1]

    one, two = ntri == 1, ntri == 2
    face_gidx = torch.cat((gid[one] * 2, torch.stack((gid[two] * 2, gid[two] * 2 + 1), -1).view(-1)))
    uvs, uv_idx = map_uv(face_gidx, tets.shape[0] * 2)
    tt = _t(TRIANGLE_TABLE)
    faces = torch.cat((torch.gather(idx_map[one], 1, tt[case[one]][:, :3]).reshape(-1, 3),
                       torch.gather(idx_map[two], 1, tt[case[two]][:, :6]).reshape(-1, 3)), 0)
    v_nrm = auto_normals(verts, faces)
    # NOTE the reference passes `faces` as the texture index too (:327 compute_tangents(..., faces, faces, faces)),
    # i.e. uvs are looked up by VERTEX id, not by uv_idx -- reproduced literally.
    v_tng = compute_tangents(verts, uvs, v_nrm, faces, faces) if faces.shape[0] > 0 else torch.zeros_like(verts)

    # ---- mSDF cut (:329-427) ---------------------------------------------------------------------
    with torch.no_grad():
        met = _t(MESH_EDGE_TABLE)
        loop3 = torch.gather(idx_map[one], 1, met[case[one]][:, [0, 1, 1, 2, 2, 0]]).view(-1, 3, 2)
        loop4 = torch.gather(idx_map[two], 1, met[case[two]][:, [0, 1, 1, 2, 2, 3, 3, 0]]).view(-1, 4, 2)
        mocc3 = (msdf_vert[loop3[:, :, 0]] > 0).long()
        mocc4 = (msdf_vert[loop4[:, :, 0]] > 0).long()

    def cut_weights(loop):
        m = msdf_vert[loop]                                              # [n, k, 2]
        ok = torch.sign(m).sum(-1).abs() != 2
        a, b = m[..., 0], -m[..., 1]
        den = a + b
        ok = ok & (den.abs() > 1e-12)
        den_safe = torch.where(ok, den, torch.ones_like(den))
        w0 = torch.where(ok, b / den_safe, torch.zeros_like(den))
        w1 = torch.where(ok, a / den_safe, torch.zeros_like(den))
        return w0, w1

    def cut(loop):
        w0, w1 = cut_weights(loop)
        bpos = verts[loop[..., 0]] * w0[..., None] + verts[loop[..., 1]] * w1[..., None]
        btng = v_tng[loop[..., 0]] * w0[..., None] + v_tng[loop[..., 1]] * w1[..., None]
        bm = msdf_vert_sg[loop[..., 0]] * w0.detach() + msdf_vert_sg[loop[..., 1]] * w1.detach()
        return bpos.reshape(-1, 3), btng.reshape(-1, 3), bm.reshape(-1)

    p3, t3, m3 = cut(loop3)
    p4, t4, m4 = cut(loop4)
    n_wt = verts.shape[0]
    verts_aug = torch.cat([verts, p3, p4], 0)
    v_tng_aug = torch.cat([v_tng, t3, t4], 0)
    msdf_aug_sg = torch.cat([msdf_vert_sg, m3, m4])

    with torch.no_grad():
        case3 = (mocc3 * _t([4, 2, 1])).sum(-1)
        case4 = (mocc4 * _t([8, 4, 2, 1])).sum(-1)
        n3 = loop3.shape[0]
        map3 = torch.cat([loop3[:, :, 0], n_wt + torch.arange(n3 * 3).view(-1, 3)], -1)
        map4 = torch.cat([loop4[:, :, 0], n_wt + n3 * 3 + torch.arange(loop4.shape[0] * 4).view(-1, 4)], -1)
        nt3 = _t(NUM_TRIANGLES_TRI)[case3]
        nt4 = _t(NUM_TRIANGLES_QUAD)[case4]
        t3t, t4t = _t(TRIANGLE_TABLE_TRI), _t(TRIANGLE_TABLE_QUAD)
        groups = []
        for k in (1, 2):
            s = nt3 == k
            groups.append(torch.gather(map3[s], 1, t3t[case3[s]][:, :3 * k]).view(-1, 3))
        for k in (1, 2, 3, 4):
            s = nt4 == k
            groups.append(torch.gather(map4[s], 1, t4t[case4[s]][:, :3 * k]).view(-1, 3))
        faces_aug = torch.cat(groups, 0)
        used = torch.zeros(verts_aug.shape[0], dtype=torch.bool)
        used[faces_aug.reshape(-1)] = True
    verts_aug = torch.where(used[:, None], verts_aug, torch.zeros_like(verts_aug))   # :423-427

    return {
        'verts': verts_aug, 'faces': faces_aug, 'v_tng': v_tng_aug,
        'n_verts_watertight': n_wt, 'vertices_watertight': verts, 'faces_watertight': faces,
        'v_tng_watertight': v_tng, 'msdf': msdf_aug_sg, 'msdf_watertight': msdf_vert_sg,
        'msdf_boundary': msdf_aug_sg[n_wt:],
        # intermediates (not returned by the reference)
        'edge_verts': ev, 'v_nrm_watertight': v_nrm, 'msdf_vert': msdf_vert,
    }


# ---- synthetic tet grids: Kuhn 6-tets-per-cube lattice -----------------------------
def kuhn_grid(n, shuffle_seed=None):
    """(n+1)^3 vertices on [-1,1]^3, 6 n^3 tets; then the reference's y -= 0.1919, *= 1.2 (hmsdf.py:210-211)."""
    g = np.arange(n + 1)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32) / n * 2 - 1
    verts[:, 1] -= np.float32(0.1919)
    verts *= np.float32(1.2)

    def vid(i, j, k):
        return (i * (n + 1) + j) * (n + 1) + k
    c = np.arange(n)
    I, J, K = np.meshgrid(c, c, c, indexing='ij')
    I, J, K = I.reshape(-1), J.reshape(-1), K.reshape(-1)
    tets = []
    import itertools
    for perm in itertools.permutations(range(3)):
        cur = [I.copy(), J.copy(), K.copy()]
        ids = [vid(*cur)]
        for ax in perm:
            cur[ax] = cur[ax] + 1
            ids.append(vid(*cur))
        tets.append(np.stack(ids, -1))
    tets = np.stack(tets, 1).reshape(-1, 4).astype(np.int64)
    if shuffle_seed is not None:
        rng = np.random.default_rng(shuffle_seed)
        tets = tets[rng.permutation(tets.shape[0])]
        for r in range(tets.shape[0]):       # also permute vertex order inside each tet (keeps it a valid tet)
            tets[r] = tets[r][rng.permutation(4)]
    return verts, tets
