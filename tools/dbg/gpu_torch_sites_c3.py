"""Which python call sites issue the torch ops left in a config-3 step?  One step under a TorchFunctionMode that logs every torch-level call whose
result lives on the GPU (fills, muls, copies, cats, adds ...) with its call site inside the package.  gpurun -- 'python tools/dbg/gpu_torch_sites_c3.py'"""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'd3human-code_amd'))
import torch
from torch.overrides import TorchFunctionMode
from d3h import scene

sc = scene.Scene(device='cuda', prefit_steps=300, res=1024, grid_n=63, n_frames=4, loss_set='full')
for _ in range(10):
    sc.step()
torch.cuda.synchronize()
log = collections.Counter()
SKIP = {'__get__', 'size', 'dim', 'is_contiguous', 'data_ptr', 'stride', '__getitem__', 'view', 'reshape', 'detach', 'shape', 'is_cuda', 'device', 'dtype',
        'numel', 'requires_grad', 'grad', 'element_size', 'untyped_storage', 'set_', 'empty', 'empty_like', 'float', 'contiguous', 'expand', 'permute',
        'unbind', 'narrow', 'squeeze', 'unsqueeze', 'transpose', 't', 'backward', 'requires_grad_', 'is_floating_point', 'new_empty', 'record_stream', '_version',
        'ndim', 'is_grad_enabled', 'storage_offset', 'nelement', 'is_leaf', 'grad_fn', 'retain_grad', 'chunk', 'split', 'flatten', 'view_as', 'type', 'to', 'item', 'tolist'}


class Log(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = getattr(func, '__name__', str(func))
        if name not in SKIP:
            st = [f for f in traceback.extract_stack(limit=12) if 'd3human-code_amd' in f.filename]
            site = f'{os.path.basename(st[-1].filename)}:{st[-1].lineno}' if st else '?'
            shp = tuple(out.shape) if torch.is_tensor(out) else ''
            log[(name, site, str(shp))] += 1
        return out


with Log():
    sc.step()
torch.cuda.synchronize()
for (name, site, shp), n in sorted(log.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f'{n:3d}  {name:22s} {site:28s} {shp}')
print('calls', sum(log.values()))
