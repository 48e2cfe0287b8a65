"""Linear blend skinning helpers with the call signatures of the reference's patched smplx
(deform/smplx_exavatar/lbs.py: lbs :156-264, batch_rodrigues :311, batch_rigid_transform :361, blend_shapes :287,
vertices2joints :267).  Thin views over d3h.smplx_pose (level-batched kinematic chain)."""
import torch

from d3h import smplx_pose as SP


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    return SP.rodrigues(rot_vecs)


def blend_shapes(betas, shape_disps):
    return torch.einsum('bl,mkl->bmk', betas, shape_disps)


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', vertices, J_regressor)


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    tree = SP.KinematicTree(parents.tolist() if torch.is_tensor(parents) else parents)
    A = tree.transforms(rot_mats, joints)
    # posed joints = translation of the un-subtracted global transforms
    posed = A[..., :3, 3] + (A[..., :3, :3] @ joints[..., None])[..., 0]
    return posed, A


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, joint_offset, locator_offset, parents, lbs_weights,
        pose2rot=True):
    """-> (verts [B,V,3], posed joints [B,J,3], A [B,J,4,4]) as the reference's patched lbs()"""
    B = max(betas.shape[0], pose.shape[0])
    v_shaped = v_template + blend_shapes(betas, shapedirs)
    J = vertices2joints(J_regressor, v_shaped)
    if joint_offset is not None:
        J = J + joint_offset
    J_skin = J + locator_offset if locator_offset is not None else J
    eye = torch.eye(3, dtype=betas.dtype, device=betas.device)
    rot = batch_rodrigues(pose.view(-1, 3)).view(B, -1, 3, 3) if pose2rot else pose.view(B, -1, 3, 3)
    feat = (rot[:, 1:] - eye).reshape(B, -1)
    v_posed = v_shaped + (feat @ posedirs).view(B, -1, 3)
    posed, A = batch_rigid_transform(rot, J_skin, parents)
    T = torch.einsum('vj,bjk->bvk', lbs_weights, A.reshape(B, -1, 16)).view(B, -1, 4, 4)
    verts = (T[..., :3, :3] @ v_posed[..., None])[..., 0] + T[..., :3, 3]
    return verts, posed, A
