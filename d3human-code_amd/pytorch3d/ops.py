"""Shim for `from pytorch3d.ops import knn_points` (deform/smplx_exavatar_deformer.py:7, geometry/hmsdf.py:44): K=1 brute force
on the HIP kernel of csrc/lbs.hip (third_parties/pytorch3d/cuda/knn.cu semantics: squared L2, first minimum wins)."""
from collections import namedtuple

import torch

from d3h import _lib as L

_KNN = namedtuple('KNN', 'dists idx knn')


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    if K != 1:
        raise NotImplementedError('d3h knn_points: K=1 only (the reference uses self.k = 1, deformer.py:40)')
    B, P = p1.shape[:2]
    idx = torch.empty(B, P, dtype=torch.int32, device=p1.device)
    dist = torch.empty(B, P, dtype=torch.float32, device=p1.device)
    for b in range(B):
        a, t = p1[b].detach().contiguous().float(), p2[b].detach().contiguous().float()
        L.check(L.lib().d3h_knn1(L.ptr(a), L.i32(P), L.ptr(t), L.i32(t.shape[0]), L.ptr(idx[b]), L.ptr(dist[b]), L.stream()), 'knn1')
    nn = None
    if return_nn:
        nn = torch.gather(p2, 1, idx.long()[..., None].expand(-1, -1, p2.shape[-1]))[:, :, None]
    return _KNN(dist[..., None], idx.long()[..., None], nn)
