"""Dataset_split of the reference (dataset/dataset_split.py:109-283): per-frame targets of one captured sequence -- colour image with
the full / garment / body mask as alpha, normal images, the (single, static) camera, the per-frame SMPL-X rows.

    ds = Dataset_split(base_dir, FLAGS, Detail=False, process_path=None, examples=None)      # the reference's signature
    loader = torch.utils.data.DataLoader(ds, batch_size=FLAGS.batch, collate_fn=ds.collate)

`base_dir` is the reference's directory layout (images/ normal/ all/ all_cloth_mask/ all_body_mask/ *.png, key.list,
smplx/merged_smplx.npz, smplx/smplx_optimized/{face_offset,joint_offset,locator_offset,shape_param}.json, smplx/cameras.npz; with
Detail=True also <process_path>/merge_body_cloth.npz and inside_body_index.npz) -- or a `MemorySource` holding the same data as arrays
(synthetic sequences, tests, tools/run_reference_train.py).  The target arithmetic is dataset/targets.py (bit-equal to the reference's
__getitem__, tests/golden/data_edges.npz); tensors go to FLAGS.device (default 'cuda', where the reference calls .cuda()).
PNG decoding uses imageio / cv2 when installed (as the reference) and PIL otherwise; images whose size differs from FLAGS.train_res are
resized with cv2.resize when available (the reference's bilinear fixed-point filter) and PIL's bilinear filter otherwise."""
import glob
import json
import os

import numpy as np
import torch

from .dataset import Dataset_people_smplx
from . import targets as _T
from .targets import get_ndc_matrix_from_ss      # noqa: F401  (dataset_split.py:57-68, same name as the reference)
from render.util import srgb_to_rgb              # noqa: F401  (dataset_split.py:24-31)

SMPLX_KEYS = ('trans', 'rhand_pose', 'jaw_pose', 'reye_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose')
_SMPLX_WIDTH = {'trans': 3, 'rhand_pose': 45, 'jaw_pose': 3, 'reye_pose': 3, 'expr': 50, 'body_pose': 63, 'root_pose': 3, 'lhand_pose': 45, 'leye_pose': 3}


def _read_png(path):
    try:
        import imageio
        if hasattr(imageio, 'imread'):
            return np.asarray(imageio.imread(path))
    except ImportError:
        pass
    from PIL import Image
    return np.asarray(Image.open(path))


def _read_color_png(path):
    """An H x W x 3 uint8 RGB image whatever the file holds -- what `cv2.imread(path, IMREAD_COLOR)` + BGR->RGB gives the reference for
    the normal maps (dataset_split.py:248-249): grey -> three equal channels, palette -> RGB, alpha dropped, 16-bit -> the high byte."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode in ('P', 'PA', 'LA', '1', 'CMYK', 'YCbCr'):
                im = im.convert('RGB')
            a = np.asarray(im)
    except ImportError:
        a = _read_png(path)
    if a.dtype == np.uint16 or a.dtype == np.int32:          # 16-bit PNG (PIL modes I;16 / I)
        a = (a.astype(np.uint32) >> 8).astype(np.uint8)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8) * 255
    if a.ndim == 2:
        a = np.stack([a, a, a], axis=-1)
    if a.ndim != 3 or a.shape[-1] not in (3, 4) or a.dtype != np.uint8:
        raise ValueError(f'{path}: cannot be read as an 8-bit colour image (shape {a.shape}, dtype {a.dtype})')
    return np.ascontiguousarray(a[..., :3])


def _resize(a, res):
    """cv2.resize(a, (res[0], res[1])) of the reference (:213,218,...): dsize is (width, height) = (train_res[0], train_res[1])"""
    w, h = int(res[0]), int(res[1])
    if a.shape[0] == h and a.shape[1] == w:
        return a
    try:
        import cv2
        if hasattr(cv2, 'resize'):
            return cv2.resize(a, (w, h))
    except ImportError:
        pass
    from PIL import Image
    return np.asarray(Image.fromarray(a).resize((w, h), Image.BILINEAR))


def read_json_files(file_path):
    with open(file_path, 'r', encoding='utf-8') as f:
        return [json.load(f)]


def load_smplx_param(root, device='cuda'):
    """dataset_split.py:82-107: merged_smplx.npz rows + the four optimised-offset json files -> dict of tensors on `device`"""
    p = dict(np.load(os.path.join(root, 'merged_smplx.npz')))
    out = {k: torch.from_numpy(p[k].astype(np.float32)).reshape(-1, _SMPLX_WIDTH[k]).to(device) for k in SMPLX_KEYS}
    for k, f in (('face_offset', 'face_offset.json'), ('joint_offset', 'joint_offset.json'), ('locator_offset', 'locator_offset.json'),
                 ('shape_param', 'shape_param.json')):
        out[k] = torch.from_numpy(np.array(read_json_files(os.path.join(root, 'smplx_optimized', f))).astype(np.float32)).to(device)
    return out


class DirectorySource:
    """the reference's on-disk layout (dataset_split.py:115-136,161-163)"""

    def __init__(self, base_dir, device):
        self.base_dir = base_dir
        with open(os.path.join(base_dir, 'key.list'), 'r') as f:
            key = [int(l.strip()) for l in f if l.strip()]
        self.begin, self.end = key[0], key[1]
        g = lambda sub: sorted(glob.glob(f'{base_dir}/{sub}/*.png'))
        self.img_lists, self.normal_lists = g('images'), g('normal')
        self.msk_lists, self.cloth_msk_lists, self.body_msk_lists = g('all'), g('all_cloth_mask'), g('all_body_mask')
        self.smplx_params = load_smplx_param(os.path.join(base_dir, 'smplx'), device)
        cam = np.load(os.path.join(base_dir, 'smplx/cameras.npz'))
        self.camera = {k: cam[k] for k in ('intrinsic', 'extrinsic', 'height', 'width')}

    def frame(self, idx):
        """-> rgb [H,W,3] u8, full / garment / body masks [H,W], normal image [H,W,3] u8 (RGB order)"""
        rgb = _read_png(self.img_lists[idx])[..., :3]
        nrm = _read_color_png(self.normal_lists[idx])             # the reference decodes with cv2 IMREAD_COLOR (always 3 x 8 bit, BGR) and converts back to RGB (:248-249)
        return rgb, _read_png(self.msk_lists[idx]), _read_png(self.cloth_msk_lists[idx]), _read_png(self.body_msk_lists[idx]), nrm

    def detail(self, process_path):
        a = np.load(os.path.join(process_path, 'merge_body_cloth.npz'))
        b = np.load(os.path.join(process_path, 'inside_body_index.npz'))
        return {'v': a['v'], 'f': a['f'], 'face_labels': a['face_labels'], 'inside_body_index': b['inside_body_index'],
                'outside_body_index': b['outside_body_index']}


class MemorySource:
    """the same data as arrays: frames = list of (rgb u8 [H,W,3], mask, garment mask, body mask [H,W], normal rgb u8 [H,W,3]); key =
    (begin, end) of key.list; smplx = dict of per-frame rows (SMPLX_KEYS) + face_offset / joint_offset / locator_offset / shape_param;
    camera = dict(intrinsic [3,3], extrinsic [4,4], height, width); detail = dict as DirectorySource.detail() or None"""

    def __init__(self, frames, key, smplx, camera, detail=None):
        self.frames, self.begin, self.end = list(frames), int(key[0]), int(key[1])
        self.smplx_params, self.camera, self._detail = smplx, camera, detail

    def frame(self, idx):
        return self.frames[idx]

    def detail(self, process_path):
        if self._detail is None:
            raise FileNotFoundError('MemorySource: no merged body + garment mesh was given (Detail=True)')
        return self._detail


class Dataset_split(Dataset_people_smplx):
    def __init__(self, base_dir, FLAGS, Detail=False, process_path=None, examples=None):
        self.FLAGS, self.examples, self.base_dir = FLAGS, examples, base_dir
        self.device = torch.device(getattr(FLAGS, 'device', 'cuda'))
        src = base_dir if hasattr(base_dir, 'frame') else DirectorySource(base_dir, self.device)
        self.source = src
        self.begin, self.end = src.begin, src.end
        self.key_frame = list(range(self.begin, self.end + 1))              # :119-126
        self.n_images = self.end - self.begin                               # :138 (one less than len(key_frame), as the reference)
        self.smplx_params = {k: (v.to(self.device) if torch.is_tensor(v) else torch.as_tensor(np.asarray(v, np.float32), device=self.device))
                             for k, v in src.smplx_params.items()}
        self.shape_param, self.face_offset = self.smplx_params['shape_param'], self.smplx_params['face_offset']
        self.joint_offset, self.locator_offset = self.smplx_params['joint_offset'], self.smplx_params['locator_offset']
        if Detail:                                                          # :149-163
            d = src.detail(process_path)
            self.v = torch.as_tensor(d['v']).to(self.device).float()
            self.f = torch.as_tensor(d['f']).to(self.device).long()
            self.face_labels = torch.as_tensor(d['face_labels']).to(self.device)
            self.inside_body_index = torch.as_tensor(d['inside_body_index']).long()
            self.outside_body_index = torch.as_tensor(d['outside_body_index']).long()
            self.cloth_index = torch.unique(self.f[self.face_labels == 1])
            self.outside_index = torch.cat((self.cloth_index.to(self.device), self.outside_body_index.to(self.device)))
        cam = src.camera                                                    # :166-204
        c = _T.camera_matrices(torch.as_tensor(np.asarray(cam['intrinsic'])), torch.as_tensor(np.asarray(cam['extrinsic'])).float(),
                               int(cam['height']), int(cam['width']))
        self.proj_mtx, self.mv, self.mvp, self.campos = c['proj'], c['mv'], c['mvp'], c['campos']
        self.w2c = torch.as_tensor(np.asarray(cam['extrinsic'])).float()
        self._cam = c

    def load_img(self, img):
        return srgb_to_rgb(torch.from_numpy(img.astype(np.float32) / 255))   # :196-199

    def __len__(self):
        return self.n_images if self.examples is None else self.examples

    def __getitem__(self, itr):
        idx = self.key_frame[itr % self.n_images]
        res = self.FLAGS.train_res
        rgb, msk, cloth, body, nrm = self.source.frame(idx)
        one = lambda m: _resize(np.where(np.asarray(m) > 0, 1, 0).astype(np.asarray(m).dtype), res)       # :216-233 (threshold, then resize)
        return _T.make_target(idx, _resize(np.asarray(rgb), res), one(msk), one(cloth), one(body), _resize(np.asarray(nrm), res), self._cam, res,
                              self.FLAGS.spp, device=self.device, smplx_params=self.smplx_params)
