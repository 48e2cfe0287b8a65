"""Small device-resident constants, uploaded once.

`torch.tensor(values, device='cuda')` copies from pageable host memory, which synchronises the stream: one such call in the middle
of the forward drains the launch queue and the rest of the pass becomes launch-bound.  Every per-iteration constant goes through this
cache instead (the tensors are shared: never write to them)."""
import torch

_CACHE = {}


def const(values, device, dtype=torch.float32):
    key = (tuple(values), str(device), dtype)
    t = _CACHE.get(key)
    if t is None:
        t = _CACHE[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return t
