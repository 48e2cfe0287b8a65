"""Checkpoint files of the reference's training loop (train.py:812-832 save, :284-331 load): `ckp/model_<it>.pt` = geometry.state_dict(),
`ckp/mtl_<it>.pt` = mat['kd_ks'].state_dict(), `ckp/smpl_<it>.pt.npz` = the nine optimised pose tensors.  Same file names, same keys
(the parameter names of geometry/hmsdf.py and render/mlptexture.py are the reference's), so checkpoints move between the two
implementations.  The HDR light probe of the reference is not written: the light is unused under bsdf = 'kd' (render/render.py:120)."""
import os

import numpy as np
import torch

POSE_KEYS = ('trans_optim', 'rhand_pose_optim', 'jaw_pose_optim', 'expr_optim', 'body_pose_optim', 'root_pose_optim', 'lhand_pose_optim',
             'leye_pose_optim', 'reye_pose_optim')


def load_filtered_state_dict(model, checkpoint_path, map_location=None):
    """train.py:284-289: entries of the file whose name and shape match the model, over the model's own state"""
    state = model.state_dict()
    loaded = torch.load(checkpoint_path, map_location=map_location)
    state.update({k: v for k, v in loaded.items() if k in state and v.size() == state[k].size()})
    return state


def save_ckp(FLAGS, save_path, it, geometry, mat):
    """train.py:812-832 (save_path is the stage directory, e.g. <out>/init)"""
    d = os.path.join(save_path, 'ckp')
    os.makedirs(d, exist_ok=True)
    with torch.no_grad():
        torch.save(geometry.state_dict(), os.path.join(d, 'model_{}.pt'.format(it)))
        torch.save(mat['kd_ks'].state_dict(), os.path.join(d, 'mtl_{}.pt'.format(it)))
    pose = {k: getattr(FLAGS, k).detach().cpu().numpy() for k in POSE_KEYS if getattr(FLAGS, k, None) is not None}
    np.savez(os.path.join(d, 'smpl_{}.pt'.format(it)), **pose)           # numpy appends .npz, as in the reference


def load_ckp(FLAGS, save_path, geometry, mat, stage, last=None, device=None):
    """train.py:292-331: `last` defaults to FLAGS.<stage>_epoch - 1"""
    if last is None:
        last = {'init': getattr(FLAGS, 'init_epoch', 1), 'split': getattr(FLAGS, 'split_epoch', 1), 'fine': getattr(FLAGS, 'fine_epoch', 1)}[stage] - 1
    d = os.path.join(save_path, stage, 'ckp')
    dev = device if device is not None else next(geometry.parameters()).device
    geometry.load_state_dict(load_filtered_state_dict(geometry, os.path.join(d, 'model_{}.pt'.format(last)), map_location=dev), strict=False)
    mat['kd_ks'].load_state_dict(load_filtered_state_dict(mat['kd_ks'], os.path.join(d, 'mtl_{}.pt'.format(last)), map_location=dev), strict=False)
    npz = np.load(os.path.join(d, 'smpl_{}.pt.npz'.format(last)))
    for k in POSE_KEYS:
        if k in npz.files:
            setattr(FLAGS, k, torch.from_numpy(npz[k]).to(dev).requires_grad_(True))
    return geometry, mat
