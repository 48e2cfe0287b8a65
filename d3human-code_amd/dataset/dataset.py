"""Dataset_people_smplx of the reference (dataset/dataset.py:146-189): the torch Dataset base class whose `collate` turns a list of
per-frame target dicts into the batch dict the tick_* functions consume (images / normals / matrices / pose rows concatenated along the
batch axis, `idx` kept as a list, resolution and spp taken from the first item)."""
import torch

_CAT_KEYS = ('img', 'img_second', 'cloth_img', 'body_img', 'all_img', 'normal', 'all_normal', 'body_normal', 'cloth_normal',
             'trans', 'rhand_pose', 'jaw_pose', 'expr', 'body_pose', 'root_pose', 'lhand_pose', 'leye_pose')


class Dataset_people_smplx(torch.utils.data.Dataset):
    """Basic dataset interface"""

    def __init__(self):
        super().__init__()

    def __len__(self):
        raise NotImplementedError

    def __getitem__(self, itr):
        raise NotImplementedError

    def collate(self, batch):
        out = {'mv': torch.cat([item['mv'] for item in batch], dim=0), 'mvp': torch.cat([item['mvp'] for item in batch], dim=0),
               'campos': torch.cat([item['campos'] for item in batch], dim=0), 'resolution': batch[0]['resolution'], 'spp': batch[0]['spp'],
               'idx': [item['idx'] for item in batch]}
        for k in _CAT_KEYS:                       # dataset.py:163-188: present in the first item -> concatenated, absent -> None
            out[k] = torch.cat([item[k] for item in batch], dim=0) if k in batch[0] else None
        return out
