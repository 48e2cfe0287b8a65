"""Target construction of dataset/dataset_split.py (Dataset_split.__init__ :181-204, __getitem__ :206-283) without the file IO.

    cam = camera_matrices(K, w2c, height, width)                 # once per sequence (the reference halves the calibration, :170-179)
    tgt = make_target(idx, rgb_u8, msk, cloth_msk, body_msk, normal_rgb_u8, cam, resolution, spp, device)

`rgb_u8`, `normal_rgb_u8`: [H,W,3] uint8 already resized to `resolution` (cv2.resize in the reference); masks: [H,W] of any dtype,
> 0 means inside.  The dict has the keys, shapes and dtypes of the reference's: images [1,H,W,4] float64 (the reference concatenates a
float32 image with a float64 mask, :236-246), normals / masks [1,H,W,C] float32, matrices [1,4,4]."""
import numpy as np
import torch

from render import util


def get_ndc_matrix_from_ss(height, width, fx, fy, cx, cy, n=0.001, f=1000.0):
    """dataset_split.py:57-68: screen-space intrinsics -> OpenGL-style NDC projection (y flipped)"""
    m = torch.zeros((4, 4))
    m[0, 0] = 2 * fx / (width - 1)
    m[0, 2] = 1 - 2 * cx / (width - 1)
    m[1, 1] = -2 * fy / (height - 1)
    m[1, 2] = 1 - 2 * cy / (height - 1)
    m[2, 2] = -(f + n) / (f - n)
    m[2, 3] = -(2 * f * n) / (f - n)
    m[3, 2] = -1.0
    return m


def camera_matrices(K, w2c, height, width, halve=True):
    """dataset_split.py:164-204: the calibration is used at half resolution (integer floor division of K and the image size), the
    world-to-camera matrix is flipped to the OpenGL convention; returns {'mv', 'mvp', 'campos', 'proj'}"""
    K = torch.as_tensor(K)
    w2c = torch.as_tensor(w2c).float()
    if halve:
        height, width = height // 2, width // 2
        fx, fy, cx, cy = K[0, 0] // 2, K[1, 1] // 2, K[0, 2] // 2, K[1, 2] // 2
    else:
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    proj = get_ndc_matrix_from_ss(height, width, fx, fy, cx, cy)
    flip = torch.tensor([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=torch.float)
    mv = flip @ w2c
    return {'mv': mv, 'mvp': proj @ mv, 'campos': torch.linalg.inv(mv)[:3, 3], 'proj': proj}


def _masked_image(rgb, msk_np):
    """:236-246: [rgb, mask] with the colour premultiplied by the mask and the alpha set to sign(mask)"""
    img = torch.from_numpy(np.concatenate((rgb.numpy(), msk_np), axis=2))
    img[:, :, :3] = img[:, :, :3] * img[:, :, 3:]
    img[:, :, 3] = torch.sign(img[:, :, 3])
    return img


def make_target(idx, rgb_u8, msk, cloth_msk, body_msk, normal_rgb_u8, cam, resolution, spp=1, device='cuda', smplx_params=None):
    rgb = util.srgb_to_rgb(torch.from_numpy(np.asarray(rgb_u8).astype(np.float32) / 255))            # load_img :196-199
    binm = lambda m: np.expand_dims((np.asarray(m) > 0).astype(np.asarray(m).dtype), axis=2).astype(float)      # :220-233
    msk_np, cloth_np, body_np = binm(msk), binm(cloth_msk), binm(body_msk)
    normal = torch.from_numpy(np.asarray(normal_rgb_u8)).float() / 255.0 * 2.0 - 1.0                   # :248-253
    normal = normal * msk_np
    out = {
        'idx': idx,
        'mv': cam['mv'][None, ...].to(device), 'mvp': cam['mvp'][None, ...].to(device), 'campos': cam['campos'][None, ...].to(device),
        'resolution': resolution, 'spp': spp,
        'all_img': _masked_image(rgb, msk_np)[None, ...].to(device),
        'cloth_img': _masked_image(rgb, cloth_np)[None, ...].to(device),
        'body_img': _masked_image(rgb, body_np)[None, ...].to(device),
        'all_normal': normal[None, ...].to(device).float(),
        'body_normal': (normal * body_np)[None, ...].to(device).float(),
        'cloth_normal': (normal * cloth_np)[None, ...].to(device).float(),
        'all_msk': torch.from_numpy(msk_np)[None, ...].to(device).float(),
        'cloth_msk': torch.from_numpy(cloth_np)[None, ...].to(device).float(),
        'body_msk': torch.from_numpy(body_np)[None, ...].to(device).float(),
    }
    if smplx_params is not None:                                                                      # :274-282
        for k in ('trans', 'rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose'):
            if k in smplx_params:
                out[k] = smplx_params[k][idx][None, ...].to(device)
    return out
