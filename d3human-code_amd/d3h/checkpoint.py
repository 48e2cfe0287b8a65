"""Checkpoint files of the reference's training loop (train.py:812-832 save, :284-331 load): `ckp/model_<it>.pt` = geometry.state_dict(),
`ckp/mtl_<it>.pt` = mat['kd_ks'].state_dict(), `ckp/smpl_<it>.pt.npz` = the nine optimised pose tensors.  Same file names, same keys
(the parameter names of geometry/hmsdf.py and render/mlptexture.py are the reference's), so checkpoints move between the two
implementations in BOTH directions: the reference's load_ckp also opens `ckp/probe_<it>.hdr` unconditionally (train.py:301,307) and
indexes all nine pose arrays (train.py:311-319), so save_ckp writes a Radiance .hdr probe (the constant 0.5 environment train.py:1747
creates and, under the forced bsdf = 'kd' of render/render.py:120, never changes -- or the light's own `base` when one is passed) and
refuses to write a pose file with a missing key."""
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


def write_hdr(path, rgb):
    """Radiance RGBE picture: rgb float [H,W,3] >= 0.  Scanlines of 8..32767 pixels are written in the new run-length format the header
    announces (2, 2, width hi, width lo, then the four channels one after the other, here as literal packets of <= 128 bytes): a flat
    scanline whose first pixel happened to be (2, 2, <128, *) -- a dark first pixel of a trained probe -- would be mis-read as one."""
    rgb = np.maximum(np.asarray(rgb, np.float32), 0.0)
    h, w, _ = rgb.shape
    mx = rgb.max(axis=-1)
    mant, expo = np.frexp(mx)                                            # mx = mant * 2^expo, mant in [0.5, 1)
    scale = np.where(mx > 1e-32, mant * 256.0 / np.maximum(mx, 1e-38), 0.0)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(mx > 1e-32, expo + 128, 0).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n' + ('-Y %d +X %d\n' % (h, w)).encode())
        if not 8 <= w <= 32767:
            f.write(out.tobytes())                                       # outside that range readers take scanlines as flat pixels
            return
        for y in range(h):
            f.write(bytes((2, 2, w >> 8, w & 255)))
            for c in range(4):
                row = out[y, :, c].tobytes()
                for x0 in range(0, w, 128):
                    chunk = row[x0:x0 + 128]
                    f.write(bytes((len(chunk),)) + chunk)


def read_hdr(path):
    """inverse of write_hdr (flat and new-RLE scanlines) -> float32 [H,W,3]; used by the tests and by load_ckp when the reference's
    render.light is not importable"""
    with open(path, 'rb') as f:
        data = f.read()
    head, _, rest = data.partition(b'\n\n')
    if not head.startswith(b'#?RADIANCE') and not head.startswith(b'#?RGBE'):
        raise ValueError(f'{path}: not a Radiance picture')
    dims, _, body = rest.partition(b'\n')
    tok = dims.split()
    if len(tok) != 4 or tok[0] != b'-Y' or tok[2] != b'+X':
        raise ValueError(f'{path}: unsupported orientation {dims!r}')
    h, w = int(tok[1]), int(tok[3])
    px = np.zeros((h, w, 4), np.uint8)
    pos = 0
    for y in range(h):
        if 8 <= w <= 32767 and body[pos] == 2 and body[pos + 1] == 2 and body[pos + 2] < 128:
            if (body[pos + 2] << 8 | body[pos + 3]) != w:
                raise ValueError(f'{path}: scanline {y} has the wrong width')
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = body[pos]; pos += 1
                    if n > 128:
                        px[y, x:x + n - 128, c] = body[pos]; pos += 1; x += n - 128
                    else:
                        px[y, x:x + n, c] = np.frombuffer(body, np.uint8, n, pos); pos += n; x += n
        else:
            px[y] = np.frombuffer(body, np.uint8, 4 * w, pos).reshape(w, 4); pos += 4 * w
    e = px[..., 3].astype(np.int32)
    f32 = np.ldexp(1.0, e - 136).astype(np.float32)
    return np.where((e > 0)[..., None], px[..., :3].astype(np.float32) * f32[..., None], 0.0).astype(np.float32)


def save_ckp(FLAGS, save_path, it, geometry, mat, lgt=None):
    """train.py:812-832 (save_path is the stage directory, e.g. <out>/init)"""
    d = os.path.join(save_path, 'ckp')
    os.makedirs(d, exist_ok=True)
    missing = [k for k in POSE_KEYS if not torch.is_tensor(getattr(FLAGS, k, None))]
    if missing:
        raise ValueError(f'save_ckp: FLAGS lacks the pose tensors {missing}; the reference loader indexes all nine (train.py:311-319)')
    with torch.no_grad():
        torch.save(geometry.state_dict(), os.path.join(d, 'model_{}.pt'.format(it)))
        torch.save(mat['kd_ks'].state_dict(), os.path.join(d, 'mtl_{}.pt'.format(it)))
    base = getattr(lgt, 'base', None)
    if torch.is_tensor(base) and base.dim() == 3:
        probe = base.detach().float().cpu().numpy()                      # latlong [H,W,3] environment of render/light.py
    else:
        res = int(getattr(FLAGS, 'probe_res', 16))
        probe = np.full((res, 2 * res, 3), 0.5, np.float32)             # create_trainable_env_rnd(res, scale=0.0, bias=0.5) (train.py:1747)
    write_hdr(os.path.join(d, 'probe_{}.hdr'.format(it)), probe)
    pose = {k: getattr(FLAGS, k).detach().cpu().numpy() for k in POSE_KEYS}
    np.savez(os.path.join(d, 'smpl_{}.pt'.format(it)), **pose)           # numpy appends .npz, as in the reference


def load_ckp(FLAGS, save_path, geometry, mat, stage, last=None, device=None):
    """train.py:292-331: `last` defaults to FLAGS.<stage>_epoch - 1.  Returns (geometry, mat, lgt) as the reference; lgt is the probe loaded
    through render.light.load_env when that module (the reference's, further down the module path) is importable, else None -- the light
    is unused under bsdf = 'kd'."""
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
    lgt = None
    probe = os.path.join(d, 'probe_{}.hdr'.format(last))
    if os.path.exists(probe):
        try:
            from render import light
        except ImportError:                  # the reference's render/light.py is not on the path: the light is unused under bsdf = 'kd'
            light = None
        if light is not None:                # a corrupt probe is an error, not a silent None
            lgt = light.load_env(probe, scale=getattr(FLAGS, 'env_scale', 1.0), res=[getattr(FLAGS, 'probe_res', 16)] * 2)
    return geometry, mat, lgt
