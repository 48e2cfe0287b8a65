"""Oracle for SMPL-X pose -> joint transforms and point skinning (torch CPU; TEST INFRASTRUCTURE).

Restates deform/smplx_exavatar/lbs.py (batch_rodrigues :311-347, batch_rigid_transform :361-413, the joint part of lbs
:216-247), deform/smplx_exavatar/body_models.py:1225-1257 (full_pose assembly; entries >= 69 zeroed) and
deform/smplx_exavatar_deformer.py (interpolate_weights :363-383, apply_lbs_inverse :385-421, lbs_forward :434-486).
Pinned by tests/golden/lbs.npz (reference functions run on a seeded miniature model).
"""
import torch
import torch.nn.functional as F


def batch_rodrigues(rv):
    angle = torch.norm(rv + 1e-8, dim=1, keepdim=True)
    d = rv / angle
    c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    z = torch.zeros_like(rx)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=rv.dtype)[None] + s * K + (1 - c) * torch.bmm(K, K)


def rigid_chain(rot, joints, parents):
    """rot [B,J,3,3], joints [B,J,3] -> posed joints [B,J,3], relative transforms A [B,J,4,4]"""
    B, J = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = rel[:, 1:] - joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=joints.dtype)
    T[:, :, :3, :3] = rot
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(chain[int(parents[i])] @ T[:, i])
    G = torch.stack(chain, 1)
    jh = F.pad(joints, [0, 1])[..., None]                       # [B,J,4,1] with 0 in the last slot
    A = G - F.pad(G @ jh, [3, 0])
    return G[:, :, :3, 3], A


def full_pose(root, body, jaw, leye, reye, lhand, rhand):
    fp = torch.cat([root.reshape(-1, 1, 3), body.reshape(-1, 21, 3), jaw.reshape(-1, 1, 3), leye.reshape(-1, 1, 3),
                    reye.reshape(-1, 1, 3), lhand.reshape(-1, 15, 3), rhand.reshape(-1, 15, 3)], 1).reshape(-1, 165).clone()
    fp[:, 69:] = 0                                              # body_models.py:1255
    return fp


def joints_from_shape(model, betas, expr, face_offset=None, joint_offset=None, locator_offset=None):
    """lbs.py:216-224: J = J_regressor (v_template [+ face_offset] + shapedirs [betas, expr]) + joint_offset (+ locator_offset)"""
    comp = torch.cat([betas, expr], -1)
    dirs = torch.cat([model['shapedirs'], model['expr_dirs']], -1)
    v = model['v_template'][None] + (face_offset if face_offset is not None else 0)
    v_shaped = v + torch.einsum('bl,mkl->bmk', comp, dirs)
    J = torch.einsum('bik,ji->bjk', v_shaped, model['J_regressor'])
    if joint_offset is not None:
        J = J + joint_offset
    if locator_offset is not None:
        J = J + locator_offset                                  # lbs.py:222-223,245-247: skin transforms use J + locator_offset
    return J


def pose_transforms(model, fp, J):
    rot = batch_rodrigues(fp.view(-1, 3)).view(fp.shape[0], -1, 3, 3)
    return rigid_chain(rot, J, model['parents'])[1]


def nearest_weights(pts, tmpl, lbs_w):
    """deformer.py:363-383 with K=1 (distance weight == 1 exactly); knn_cpu.cpp:13-69 order, first minimum wins"""
    d = pts[:, None, :] - tmpl[None]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    idx = torch.argmin(d2, 1)
    return idx, lbs_w[idx]


def knn1(pts, tmpl, return_dist=False):
    """K=1 knn_points (knn_cpu.cpp:13-69): squared L2 accumulated x, y, z in that order; the sequential scan keeps the first minimum"""
    d = pts[:, None, :] - tmpl[None]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    d2 = torch.where(torch.isnan(d2), torch.full_like(d2, float('inf')), d2)     # `d < best` is false for NaN: such entries never win
    idx = torch.argmin(d2, 1)
    return (idx, d2.gather(1, idx[:, None])[:, 0]) if return_dist else idx


def blend_apply(pts, A, w, inverse):
    """deformer.py:385-421"""
    M = torch.einsum('pj,jab->pab', w, A)
    if inverse:
        M = torch.inverse(M)
    ph = torch.cat([pts, torch.ones_like(pts[:, :1])], 1)[..., None]
    return (M @ ph)[:, :3, 0]


def lbs_forward(pts, tmpl, lbs_w, A0, A, trans):
    """deformer.py:434-486 for one frame: pts [P,3], A0/A [J,4,4], trans [3]"""
    idx, w = nearest_weights(pts, tmpl, lbs_w)
    can = blend_apply(pts, A0, w, True)
    return blend_apply(can, A, w, False) + trans[None], idx, can
