"""Oracle for rasterize / interpolate / antialias / texture (numpy float32 for the discrete decisions, torch for the
differentiable arithmetic).  TEST INFRASTRUCTURE.

PARITY UNPINNED: these four ops replace nvdiffrast (README.md:29, un-vendored, unpinned; call sites render/render.py:37,
72,102,381,400-403).  No source, test or golden vector for them exists under the reference tree, so this file restates
the library's published behaviour and is the specification the HIP kernels are checked against;
it cannot be checked against nvdiffrast itself in this environment.
"""
import numpy as np
import torch

from . import fl

f32 = np.float32


W_EPS = f32(1e-8)


def _nudge_w(w):
    """|w| < 1e-8 -> +-1e-8 (a vertex exactly on the camera plane has no projection; -0 / +0 count as +)"""
    return np.where(np.abs(w) < W_EPS, np.where(w < 0, -W_EPS, W_EPS), w).astype(f32)


def _setup(pos_b, tri):
    """pos_b [V,4] float32 numpy -> per-triangle NDC X,Y [F,3], q=1/w, z/w, ok (all w > 1e-8), cross (some but not all w > 1e-8)"""
    p = pos_b[tri]                                   # [F,3,4]
    w = p[..., 3]
    front = w > W_EPS
    ok = np.all(front, axis=1)
    cross = np.any(front, axis=1) & ~ok
    with np.errstate(divide='ignore', invalid='ignore'):
        q = f32(1.0) / _nudge_w(w)
    X, Y, ZW = p[..., 0] * q, p[..., 1] * q, p[..., 2] * q
    return X, Y, q, ZW, ok, cross


def _order_key(z):
    u = np.asarray(z, f32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint64)


def rasterize_ids(pos, tri, H, W):
    """discrete part: winning triangle id+1 per pixel [B,H,W] (0 = empty); nearest z/w wins, ties -> lower id"""
    B = pos.shape[0]
    out = np.zeros((B, H, W), np.int64)
    sxW, syH = f32(2.0) / f32(W), f32(2.0) / f32(H)
    for b in range(B):
        X, Y, q, ZW, ok, cross = _setup(pos[b], tri)
        key = np.full((H, W), np.uint64(0xFFFFFFFFFFFFFFFF))
        for f in range(tri.shape[0]):
            if not ok[f] and not cross[f]:
                continue                               # entirely behind the camera plane
            x, y = X[f], Y[f]
            if cross[f]:
                # Near-plane crossing (some w <= 0): no explicit clipping -- the homogeneous form of the edge functions.  With
                # a_k the NDC edge functions and n_k = a_k q_k, the perspective-correct barycentrics are n_k / S (S = sum n_k) whatever
                # the signs of the q_k (the common factor q0 q1 q2 cancels), the interpolated w is s / S (s = sum a_k) and z/w is
                # sum(a_k zw_k) / s.  A pixel is covered iff all barycentrics are >= 0 and the interpolated w is > 0; the depth range
                # test [-1, 1] then removes what lies in front of the near plane.  Bounding box: the whole frame.
                x0, x1, y0, y1 = 0, W - 1, 0, H - 1
            else:
                area = (x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0])
                if area == 0:
                    continue
                x0 = int(max(0.0, np.ceil((x.min() + f32(1)) * f32(0.5) * f32(W) - f32(0.5))))
                x1 = int(min(W - 1.0, np.floor((x.max() + f32(1)) * f32(0.5) * f32(W) - f32(0.5))))
                y0 = int(max(0.0, np.ceil((y.min() + f32(1)) * f32(0.5) * f32(H) - f32(0.5))))
                y1 = int(min(H - 1.0, np.floor((y.max() + f32(1)) * f32(0.5) * f32(H) - f32(0.5))))
            if x1 < x0 or y1 < y0:
                continue
            px, py = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
            fx = (px.astype(f32) + f32(0.5)) * sxW - f32(1)
            fy = (py.astype(f32) + f32(0.5)) * syH - f32(1)
            dx = [x[k] - fx for k in range(3)]
            dy = [y[k] - fy for k in range(3)]
            a0 = dx[1] * dy[2] - dy[1] * dx[2]
            a1 = dx[2] * dy[0] - dy[2] * dx[0]
            a2 = dx[0] * dy[1] - dy[0] * dx[1]
            s = a0 + a1 + a2
            if cross[f]:
                with np.errstate(over='ignore', invalid='ignore'):
                    n0, n1, n2 = a0 * q[f, 0], a1 * q[f, 1], a2 * q[f, 2]
                    S = (n0 + n1) + n2
                    inside = (n0 * S >= 0) & (n1 * S >= 0) & (n2 * S >= 0) & (S != 0) & (s * S > 0) & np.isfinite(S)
            else:
                inside = ((a0 >= 0) & (a1 >= 0) & (a2 >= 0)) if area > 0 else ((a0 <= 0) & (a1 <= 0) & (a2 <= 0))
            inside &= s != 0
            with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
                zw = ((a0 * ZW[f, 0] + a1 * ZW[f, 1]) + a2 * ZW[f, 2]) * (f32(1) / s)
            inside &= (zw >= -1) & (zw <= 1)
            k = (_order_key(zw) << np.uint64(32)) | np.uint64(f + 1)
            sub = key[y0:y1 + 1, x0:x1 + 1]
            upd = inside & (k < sub)
            sub[upd] = k[upd]
        idm = (key & np.uint64(0xFFFFFFFF)).astype(np.int64)
        idm[key == np.uint64(0xFFFFFFFFFFFFFFFF)] = 0
        out[b] = idm
    return out


def rasterize(pos, tri, H, W, ids=None):
    """pos: torch [B,V,4] (may require grad), tri: LongTensor [F,3] -> rast [B,H,W,4], db [B,H,W,4] (torch).
    `ids` [B,H,W] (triangle id + 1, 0 = empty): use these winners instead of running the discrete pass (tests hand in another
    rasteriser's decisions to compare everything downstream of them strictly)"""
    ids = torch.from_numpy(rasterize_ids(pos.detach().numpy().astype(f32), tri.numpy(), H, W)) if ids is None else ids.long()
    B = pos.shape[0]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    fx = ((fl(xs) + 0.5) * (2.0 / W) - 1.0)[None].expand(B, -1, -1)
    fy = ((fl(ys) + 0.5) * (2.0 / H) - 1.0)[None].expand(B, -1, -1)
    cov = ids > 0
    f = (ids - 1).clamp(min=0)
    bi = torch.arange(B)[:, None, None].expand(-1, H, W)
    P = pos[bi[..., None], tri[f]]                          # [B,H,W,3,4]
    wv = P[..., 3]
    wv = torch.where(wv.abs() < 1e-8, torch.where(wv < 0, -torch.ones_like(wv), torch.ones_like(wv)) * 1e-8, wv)     # as _nudge_w
    q = 1.0 / wv
    X, Y, ZW = P[..., 0] * q, P[..., 1] * q, P[..., 2] * q
    dx, dy = X - fx[..., None], Y - fy[..., None]
    a = torch.stack([dx[..., 1] * dy[..., 2] - dy[..., 1] * dx[..., 2],
                     dx[..., 2] * dy[..., 0] - dy[..., 2] * dx[..., 0],
                     dx[..., 0] * dy[..., 1] - dy[..., 0] * dx[..., 1]], -1)
    s = a.sum(-1)
    n = a * q
    S = n.sum(-1)
    # uncovered pixels evaluate triangle 0 as a placeholder: if that triangle is degenerate there (S = 0), 0 / 0 is masked in the value by the
    # `where` below but NOT in the gradient (0 * nan): every vertex of triangle 0 then gets a NaN gradient.  (Found at full size, round 4: the
    # first face of a fitted mesh had collapsed; the HIP rasteriser never touches uncovered pixels.)
    S = torch.where(cov, S, torch.ones_like(S))
    s = torch.where(cov, s, torch.ones_like(s))
    u, v = n[..., 0] / S, n[..., 1] / S
    zw = (a * ZW).sum(-1) / s
    dax = torch.stack([Y[..., 1] - Y[..., 2], Y[..., 2] - Y[..., 0], Y[..., 0] - Y[..., 1]], -1) * q
    day = torch.stack([X[..., 2] - X[..., 1], X[..., 0] - X[..., 2], X[..., 1] - X[..., 0]], -1) * q
    dSx, dSy = dax.sum(-1), day.sum(-1)
    sx, sy = 2.0 / W, 2.0 / H
    db = torch.stack([(dax[..., 0] - u * dSx) / S * sx, (day[..., 0] - u * dSy) / S * sy,
                      (dax[..., 1] - v * dSx) / S * sx, (day[..., 1] - v * dSy) / S * sy], -1)
    rast = torch.stack([u, v, zw.detach(), fl(ids)], -1)
    z = torch.zeros_like(rast)
    return torch.where(cov[..., None], rast, z), torch.where(cov[..., None], db.detach(), z)


def interpolate(attr, rast, tri, rast_db=None):
    """attr [B or 1,V,A] -> out [B,H,W,A] (+ pixel derivatives [B,H,W,2A] when rast_db is given)"""
    B, H, W = rast.shape[:3]
    ids = rast[..., 3].long()
    cov = ids > 0
    f = (ids - 1).clamp(min=0)
    bi = torch.arange(B)[:, None, None].expand(-1, H, W) if attr.shape[0] > 1 else torch.zeros(B, H, W, dtype=torch.long)
    a = attr[bi[..., None], tri[f]]                         # [B,H,W,3,A]
    u, v = rast[..., 0:1], rast[..., 1:2]
    out = u * a[..., 0, :] + v * a[..., 1, :] + (1 - u - v) * a[..., 2, :]
    out = torch.where(cov[..., None], out, torch.zeros_like(out))
    if rast_db is None:
        return out, None
    e0, e1 = a[..., 0, :] - a[..., 2, :], a[..., 1, :] - a[..., 2, :]
    dX = rast_db[..., 0:1] * e0 + rast_db[..., 2:3] * e1
    dY = rast_db[..., 1:2] * e0 + rast_db[..., 3:4] * e1
    da = torch.stack([dX, dY], -1).reshape(B, H, W, -1)
    return out, torch.where(cov[..., None], da, torch.zeros_like(da))


def texture(tex, uv):
    """bilinear, clamp; tex [B or 1,TH,TW,C], uv [B,H,W,2]"""
    B = uv.shape[0]
    TH, TW = tex.shape[1:3]
    x = uv[..., 0] * TW - 0.5
    y = uv[..., 1] * TH - 0.5
    xf, yf = torch.floor(x), torch.floor(y)
    fx, fy = (x - xf)[..., None], (y - yf)[..., None]
    x0, x1 = xf.long().clamp(0, TW - 1), (xf.long() + 1).clamp(0, TW - 1)
    y0, y1 = yf.long().clamp(0, TH - 1), (yf.long() + 1).clamp(0, TH - 1)
    bi = torch.arange(B)[:, None, None].expand_as(x0) if tex.shape[0] > 1 else torch.zeros_like(x0)
    t = lambda yy, xx: tex[bi, yy, xx]
    return (t(y0, x0) * (1 - fx) + t(y0, x1) * fx) * (1 - fy) + (t(y1, x0) * (1 - fx) + t(y1, x1) * fx) * fy


def _edge_map(tri):
    m = {}
    for f, t in enumerate(tri.tolist()):
        for e in range(3):
            va, vb, vo = t[e], t[(e + 1) % 3], t[(e + 2) % 3]
            if va == vb:
                continue
            m.setdefault((min(va, vb), max(va, vb)), []).append(vo)
    return m


def antialias(color, rast, pos, tri):
    """color [B,H,W,C], rast [B,H,W,4], pos [B or 1,V,4] (torch, differentiable in color and pos)"""
    B, H, W, C = color.shape
    emap = _edge_map(tri)
    rn = rast.detach().numpy()
    tri_l = tri.tolist()
    items = []      # (b, dst, src, sign, horizontal, pi, po, va, vb)
    posn = pos.detach().numpy().astype(f32)
    for b in range(B):
        pb = posn[b if pos.shape[0] > 1 else 0]
        for y in range(H):
            for x in range(W):
                for dirx in (1, 0):
                    x1, y1 = x + dirx, y + (1 - dirx)
                    if x1 >= W or y1 >= H:
                        continue
                    t0, t1 = int(rn[b, y, x, 3]), int(rn[b, y1, x1, 3])
                    if t0 == t1:
                        continue
                    first = (t1 == 0) or (t0 != 0 and rn[b, y, x, 2] < rn[b, y1, x1, 2])
                    tf = t0 if first else t1
                    (xi, yi), (xo, yo) = ((x, y), (x1, y1)) if first else ((x1, y1), (x, y))
                    cxi, cyi, cxo, cyo = f32(xi + 0.5), f32(yi + 0.5), f32(xo + 0.5), f32(yo + 0.5)
                    vid = tri_l[tf - 1]
                    P = pb[vid]
                    if not np.all(P[:, 3] > W_EPS):
                        continue          # a triangle that crosses the camera plane has no screen-space silhouette edges: not blended
                    qq = f32(1) / P[:, 3]
                    sx = (P[:, 0] * qq * f32(0.5) + f32(0.5)) * f32(W)
                    sy = (P[:, 1] * qq * f32(0.5) + f32(0.5)) * f32(H)
                    for e in range(3):
                        ia, ib, io = e, (e + 1) % 3, (e + 2) % 3
                        xa, ya, xb, yb = sx[ia], sy[ia], sx[ib], sy[ib]
                        if dirx:
                            if not (min(ya, yb) <= cyi <= max(ya, yb)) or ya == yb:
                                continue
                            if abs(yb - ya) < abs(xb - xa):      # horizontal pairs only see edges closer to vertical
                                continue
                            tt = (cyi - ya) / (yb - ya)
                            d = ((xa + tt * (xb - xa)) - cxi) / (cxo - cxi)
                        else:
                            if not (min(xa, xb) <= cxi <= max(xa, xb)) or xa == xb:
                                continue
                            if abs(xb - xa) < abs(yb - ya):      # vertical pairs only see edges closer to horizontal
                                continue
                            tt = (cxi - xa) / (xb - xa)
                            d = ((ya + tt * (yb - ya)) - cyi) / (cyo - cyi)
                        if not (0 <= d <= 1):
                            continue
                        opp = emap.get((min(vid[ia], vid[ib]), max(vid[ia], vid[ib])), [])
                        other = -1
                        if len(opp) >= 2:
                            other = opp[1] if opp[0] == vid[io] else opp[0]
                        if other >= 0 and pb[other, 3] > 1e-8:
                            qo = f32(1) / pb[other, 3]
                            ox = (pb[other, 0] * qo * f32(0.5) + f32(0.5)) * f32(W)
                            oy = (pb[other, 1] * qo * f32(0.5) + f32(0.5)) * f32(H)
                            sc = (xb - xa) * (sy[io] - ya) - (yb - ya) * (sx[io] - xa)
                            so = (xb - xa) * (oy - ya) - (yb - ya) * (ox - xa)
                            if sc * so < 0:
                                continue
                        items.append((b, yi * W + xi, yo * W + xo, dirx, vid[ia], vid[ib], float(cxi), float(cyi), float(cxo), float(cyo)))
                        break
    out = color.reshape(B, H * W, C).clone()
    if not items:
        return out.reshape(B, H, W, C)
    it = np.array([[i[0], i[1], i[2], i[3], i[4], i[5]] for i in items], np.int64)
    cen = torch.tensor([[i[6], i[7], i[8], i[9]] for i in items], dtype=torch.float32)
    bb, pi, po, dirx, va, vb = [torch.from_numpy(it[:, k]) for k in range(6)]
    pb_idx = bb if pos.shape[0] > 1 else torch.zeros_like(bb)
    Pa, Pb = pos[pb_idx, va], pos[pb_idx, vb]
    sxa = (Pa[:, 0] / Pa[:, 3] * 0.5 + 0.5) * W
    sya = (Pa[:, 1] / Pa[:, 3] * 0.5 + 0.5) * H
    sxb = (Pb[:, 0] / Pb[:, 3] * 0.5 + 0.5) * W
    syb = (Pb[:, 1] / Pb[:, 3] * 0.5 + 0.5) * H
    hor = dirx.bool()
    tt_h = (cen[:, 1] - sya) / torch.where(hor, syb - sya, torch.ones_like(sya))
    d_h = ((sxa + tt_h * (sxb - sxa)) - cen[:, 0]) / torch.where(hor, cen[:, 2] - cen[:, 0], torch.ones_like(sya))
    tt_v = (cen[:, 0] - sxa) / torch.where(~hor, sxb - sxa, torch.ones_like(sya))
    d_v = ((sya + tt_v * (syb - sya)) - cen[:, 1]) / torch.where(~hor, cen[:, 3] - cen[:, 1], torch.ones_like(sya))
    d = torch.where(hor, d_h, d_v)
    alpha = d - 0.5
    pos_side = alpha >= 0
    dst = torch.where(pos_side, po, pi)
    src = torch.where(pos_side, pi, po)
    cin = color.reshape(B, H * W, C)
    delta = alpha.abs()[:, None] * (cin[bb, src] - cin[bb, dst])
    out = out.index_put((bb, dst), delta, accumulate=True)
    return out.reshape(B, H, W, C)
