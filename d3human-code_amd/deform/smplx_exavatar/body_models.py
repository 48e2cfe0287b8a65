"""Minimal SMPL-X layer with the interface the deformer uses (deform/smplx_exavatar/body_models.py:1125-1315):
forward(...) -> (output with .vertices/.joints, A).  Loads the official SMPLX_{GENDER}.npz when the licence-gated file
is present under `model_path`, or is built from a dict (d3h.synth.make_body_model) -- no SMPL/SMPLH/MANO/FLAME layers,
landmarks or PCA hands: the hot path consumes only A (and, once at initialisation, the posed template vertices)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from d3h import smplx_pose as SP
from . import lbs as LBS


class SMPLX(torch.nn.Module):
    NUM_BODY_JOINTS = 21
    NUM_JOINTS = 55

    def __init__(self, model_path=None, model_type='smplx', gender='neutral', num_betas=100, num_expression_coeffs=50,
                 model_dict=None, dtype=torch.float32, **kwargs):
        super().__init__()
        if model_dict is None:
            fn = model_path if model_path and model_path.endswith('.npz') else os.path.join(model_path or '.', 'smplx', f'SMPLX_{gender.upper()}.npz')
            if not os.path.exists(fn):
                fn2 = os.path.join(model_path or '.', f'SMPLX_{gender.upper()}.npz')
                fn = fn2 if os.path.exists(fn2) else fn
            if not os.path.exists(fn):
                raise FileNotFoundError(f'SMPL-X model file not found: {fn} (licence-gated; pass model_dict=d3h.synth.make_body_model() '
                                        f'for the synthetic body)')
            d = np.load(fn, allow_pickle=True)
            sd = np.asarray(d['shapedirs'], np.float32)
            model_dict = {'v_template': np.asarray(d['v_template'], np.float32), 'weights': np.asarray(d['weights'], np.float32),
                          'J_regressor': np.asarray(d['J_regressor'], np.float32), 'shapedirs': sd[:, :, :num_betas],
                          'expr_dirs': sd[:, :, 300:300 + num_expression_coeffs],
                          'posedirs': np.reshape(np.asarray(d['posedirs'], np.float32), [-1, np.asarray(d['posedirs']).shape[-1]]).T,
                          'parents': np.asarray(d['kintree_table'][0], np.int64), 'f': np.asarray(d['f'], np.int64)}
        t = lambda k: torch.as_tensor(np.asarray(model_dict[k]), dtype=dtype)
        for k in ('v_template', 'J_regressor', 'shapedirs', 'expr_dirs', 'posedirs'):
            self.register_buffer(k, t(k))
        self.register_buffer('lbs_weights', t('weights'))
        par = [int(p) for p in np.asarray(model_dict['parents'])]
        par[0] = -1
        self.register_buffer('parents', torch.tensor(par, dtype=torch.long))
        self.tree = SP.KinematicTree(par)
        self.register_buffer('parents32', torch.tensor([max(p_, 0) for p_ in par], dtype=torch.int32))
        # the joint regressor folded through the template and the blend-shape bases once: J = J0 + JD [betas, expr] (lbs.py:216-218 is
        # linear in them); a face_offset adds J_regressor face_offset per call
        with torch.no_grad():
            self.register_buffer('J0', torch.einsum('ik,ji->jk', t('v_template'), t('J_regressor')))
            self.register_buffer('JD', torch.einsum('mkl,jm->jkl', torch.cat([t('shapedirs'), t('expr_dirs')], -1), t('J_regressor')))
        f = model_dict.get('f')
        self.faces = None if f is None else np.asarray(f)
        self.faces_tensor = None if f is None else torch.as_tensor(np.asarray(f), dtype=torch.long)

    def joints(self, betas, expression, face_offset=None, joint_offset=None, locator_offset=None):
        comp = torch.cat([betas, expression.expand(betas.shape[0], -1) if expression.shape[0] != betas.shape[0] else expression], -1)
        dirs = torch.cat([self.shapedirs, self.expr_dirs], -1)
        v = self.v_template if face_offset is None else self.v_template + face_offset
        v_shaped = v + torch.einsum('bl,mkl->bmk', comp, dirs)
        J = torch.einsum('bik,ji->bjk', v_shaped, self.J_regressor)
        if joint_offset is not None:
            J = J + joint_offset
        if locator_offset is not None:
            J = J + locator_offset
        return J, v_shaped

    def joints_fast(self, betas, expression, face_offset=None, joint_offset=None, locator_offset=None):
        """== joints()[0] through the pre-regressed bases (no pass over the 10 475 template vertices)"""
        comp = torch.cat([betas, expression.expand(betas.shape[0], -1) if expression.shape[0] != betas.shape[0] else expression], -1)
        J = self.J0[None] + torch.einsum('jkl,bl->bjk', self.JD, comp)
        if face_offset is not None:
            J = J + torch.einsum('bik,ji->bjk', face_offset.expand(J.shape[0], -1, -1) if face_offset.dim() == 3 else face_offset[None], self.J_regressor)
        if joint_offset is not None:
            J = J + joint_offset
        if locator_offset is not None:
            J = J + locator_offset
        return J

    def transforms(self, betas, global_orient, body_pose, jaw_pose, expression, face_offset=None, joint_offset=None,
                   locator_offset=None):
        """A [B,55,4,4] only -- what lbs_forward needs (skips vertex skinning and landmarks): rest joints from the pre-regressed bases,
        then ONE kernel for Rodrigues + the kinematic chain + the rest-pose removal (csrc/smplx_pose.hip)"""
        B = body_pose.reshape(-1, 63).shape[0]
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        J = self.joints_fast(betas, expression.reshape(B, -1), face_offset, joint_offset, locator_offset)
        fp = SP.assemble_full_pose(global_orient, body_pose, jaw_pose, None, None, None, None)
        return SP.pose_transforms(fp, J, self.parents32)

    def transforms_reference(self, betas, global_orient, body_pose, jaw_pose, expression, face_offset=None, joint_offset=None,
                             locator_offset=None):
        """the level-batched torch formulation of round 1 (kept as the in-package cross-check of the kernel)"""
        B = body_pose.reshape(-1, 63).shape[0]
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        J, _ = self.joints(betas, expression.reshape(B, -1), face_offset, joint_offset, locator_offset)
        fp = SP.assemble_full_pose(global_orient, body_pose, jaw_pose, None, None, None, None)
        rot = SP.rodrigues(fp.view(-1, 3)).view(B, -1, 3, 3)
        return self.tree.transforms(rot, J)

    def forward(self, betas=None, global_orient=None, body_pose=None, left_hand_pose=None, right_hand_pose=None, transl=None,
                expression=None, jaw_pose=None, leye_pose=None, reye_pose=None, face_offset=None, joint_offset=None,
                locator_offset=None, return_verts=True, pose2rot=True, **kwargs):
        B = body_pose.reshape(-1, 63).shape[0]
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        expression = expression.reshape(-1, self.expr_dirs.shape[-1])
        if expression.shape[0] != B:
            expression = expression.expand(B, -1)
        fp = SP.assemble_full_pose(global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose)
        comp = torch.cat([betas, expression], -1)
        dirs = torch.cat([self.shapedirs, self.expr_dirs], -1)
        vt = self.v_template if face_offset is None else self.v_template + face_offset
        verts, joints, A = LBS.lbs(comp, fp, vt, dirs, self.posedirs, self.J_regressor, joint_offset, locator_offset, self.parents,
                                   self.lbs_weights, pose2rot=True)
        if transl is not None:
            verts = verts + transl[:, None]
            joints = joints + transl[:, None]
        return SimpleNamespace(vertices=verts, joints=joints, betas=betas, expression=expression, global_orient=global_orient,
                               body_pose=body_pose, transl=transl), A


def create(model_path, model_type='smplx', **kwargs):
    return SMPLX(model_path, **kwargs)
