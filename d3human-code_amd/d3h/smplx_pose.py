"""SMPL-X pose -> per-joint skinning transforms A[B,55,4,4] (the only SMPL-X output the hot path consumes,
deform/smplx_exavatar_deformer.py:456,476).

Follows deform/smplx_exavatar/lbs.py (joints :216-224, batch_rodrigues :311-347, batch_rigid_transform :361-413) and the
full-pose assembly of body_models.py:1225-1257, but walks the kinematic tree level by level (all joints of one depth in
one batched matmul: 10 steps instead of the reference's 54 sequential 4x4 products) and skips the vertex skinning /
landmark code whose results the reference discards.  Differentiable through torch autograd (tiny tensors: plumbing).
"""
import torch
import torch.nn.functional as F


def rodrigues(rv):
    """axis-angle [N,3] -> rotation matrices [N,3,3] (lbs.py:311-347, incl. the +1e-8 inside the norm)"""
    angle = torch.norm(rv + 1e-8, dim=1, keepdim=True)
    axis = rv / angle
    s, c = torch.sin(angle)[..., None], torch.cos(angle)[..., None]
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=rv.dtype, device=rv.device)[None] + s * K + (1 - c) * (K @ K)


class KinematicTree:
    def __init__(self, parents):
        parents = [int(p) for p in parents]
        self.parents = parents
        depth = [0] * len(parents)
        for j in range(1, len(parents)):
            depth[j] = depth[parents[j]] + 1
        self.levels = []
        for d in range(1, max(depth) + 1):
            js = [j for j in range(len(parents)) if depth[j] == d]
            self.levels.append((js, [parents[j] for j in js]))

    def transforms(self, rot, joints):
        """rot [B,J,3,3], joints [B,J,3] -> A [B,J,4,4] = G_j with the rest joint subtracted (lbs.py:407-411)"""
        B, J = joints.shape[:2]
        par = torch.as_tensor(self.parents[1:], device=joints.device)
        rel = torch.cat([joints[:, :1], joints[:, 1:] - joints[:, par]], 1)
        T = torch.cat([torch.cat([rot, rel[..., None]], -1),
                       torch.tensor([0., 0., 0., 1.], dtype=rot.dtype, device=rot.device).expand(B, J, 1, 4)], -2)
        G = [None] * J
        G[0] = T[:, 0]
        for js, ps in self.levels:
            Gp = torch.stack([G[p] for p in ps], 1)
            Gc = Gp @ T[:, js]
            for k, j in enumerate(js):
                G[j] = Gc[:, k]
        G = torch.stack(G, 1)
        t = G[..., :3, 3] - (G[..., :3, :3] @ joints[..., None])[..., 0]
        return torch.cat([torch.cat([G[..., :3, :3], t[..., None]], -1), G[..., 3:, :]], -2)


def assemble_full_pose(root, body, jaw, leye, reye, lhand, rhand):
    B = body.reshape(-1, 63).shape[0]
    fp = torch.cat([root.reshape(B, 3), body.reshape(B, 63), jaw.reshape(B, 3)], 1)
    # body_models.py:1255 zeroes entries >= 69 (eyes + both hands): they never influence A
    return torch.cat([fp, fp.new_zeros(B, 165 - 69)], 1)


class _PoseFn(torch.autograd.Function):
    """(full pose [B,J,3], rest joints [B or 1,J,3]) -> A [B,J,4,4] on the single-launch kernels of csrc/smplx_pose.hip"""

    @staticmethod
    def forward(ctx, fp, joints, parents32):
        from . import _lib as L
        fpc, jc = fp.contiguous().float(), joints.contiguous().float()
        B, J = fpc.shape[0], fpc.shape[1]
        A = torch.empty(B, J, 4, 4, dtype=torch.float32, device=fp.device)
        G = torch.empty(B, J, 12, dtype=torch.float32, device=fp.device)
        jbs = 0 if jc.shape[0] == 1 else J * 3
        L.check(L.lib().d3h_smplx_pose_fwd(L.ptr(fpc), L.ptr(jc), L.i32(jbs), L.ptr(parents32), L.i32(J), L.i32(B), L.ptr(A), L.ptr(G), L.stream()),
                'smplx_pose_fwd')
        ctx.save_for_backward(fpc, jc, parents32, G)
        ctx.jshape = joints.shape
        return A

    @staticmethod
    def backward(ctx, dA):
        from . import _lib as L
        fpc, jc, parents32, G = ctx.saved_tensors
        B, J = fpc.shape[0], fpc.shape[1]
        d_fp = torch.empty_like(fpc)
        d_J = torch.empty(B, J, 3, dtype=torch.float32, device=fpc.device) if ctx.needs_input_grad[1] else None
        jbs = 0 if jc.shape[0] == 1 else J * 3
        L.check(L.lib().d3h_smplx_pose_bwd(L.ptr(fpc), L.ptr(jc), L.i32(jbs), L.ptr(parents32), L.i32(J), L.i32(B), L.ptr(G), L.ptr(dA.contiguous().float()),
                                           L.ptr(d_fp), L.ptr(d_J), L.stream()), 'smplx_pose_bwd')
        if d_J is not None and tuple(d_J.shape) != tuple(ctx.jshape):
            d_J = d_J.sum_to_size(ctx.jshape)
        return d_fp, d_J, None


def pose_transforms(full_pose, joints, parents32):
    """full_pose [B,165] (or [B,55,3]) axis-angle, joints [B or 1,55,3] -> A [B,55,4,4] (lbs.py:311-413), one kernel launch"""
    B = full_pose.shape[0]
    return _PoseFn.apply(full_pose.reshape(B, -1, 3), joints, parents32)
