"""Import harness for the read-only reference at /root/reference (dev container only).

Used ONLY by tools/gen_golden.py to (i) validate the CPU oracle op-by-op and (ii) emit the small
fixtures under tests/golden/.  Nothing here is imported by the product, by bench.py or by the
GPU tests; /root/reference does not exist on the GPU box.

The reference hard-codes device='cuda' (e.g. geometry/gshell_tets.py:108,283); a TorchFunctionMode
rewrites that to 'cpu'.  Un-vendored third-party modules are stubbed.
"""
import sys, types
import torch
from torch.overrides import TorchFunctionMode

REF = '/root/reference'


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if 'cuda' in str(kwargs.get('device', '')):
            kwargs['device'] = 'cpu'
        name = getattr(func, '__name__', '')
        if name == 'cuda':
            return args[0]
        if name == 'is_cuda':
            return False
        return func(*args, **kwargs)


def stub(name, **attrs):
    parts = name.split('.')
    for i in range(1, len(parts) + 1):
        n = '.'.join(parts[:i])
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
            if i > 1:
                setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
    for k, v in attrs.items():
        setattr(sys.modules[name], k, v)
    return sys.modules[name]


def knn_points_cpu(p1, p2, K=1, return_nn=False, **kw):
    """== third_parties/pytorch3d/cuda/knn_cpu.cpp:13-69 for K=1: squared L2, first minimum wins."""
    from collections import namedtuple
    d = torch.cdist(p1.double(), p2.double()).pow(2).float()
    dist, idx = torch.topk(d, K, dim=-1, largest=False)
    knn = None
    if return_nn:       # [B,N,K,3] gathered neighbours (differentiable w.r.t. p2), as pytorch3d's knn_gather
        knn = torch.stack([p2[b][idx[b]] for b in range(p2.shape[0])])
    return namedtuple('KNN', 'dists idx knn')(dist, idx, knn)


_installed = False


def install():
    global _installed
    if _installed:
        return
    sys.dont_write_bytecode = True
    # drop any repo-side namespace packages of the same names
    sys.path = [p for p in sys.path if 'd3human-code_amd' not in p]
    sys.path.insert(0, REF)
    for n in ['nvdiffrast.torch', 'imageio', 'tinycudann', 'render.optixutils', 'pytorch3d.io',
              'pytorch3d.ops', 'kaolin', 'pysdf', 'trimesh', 'torchvision', 'torchvision.models', 'cv2']:
        if n == 'render.optixutils':
            # must be pre-stubbed: its ops.py JIT-builds OptiX at import (render/optixutils/ops.py:18-75)
            import importlib
            import render  # namespace pkg from the reference
            m = types.ModuleType('render.optixutils')
            m.__path__ = []
            sys.modules['render.optixutils'] = m
            setattr(render, 'optixutils', m)
        else:
            stub(n)
    sys.modules['pytorch3d.ops'].knn_points = knn_points_cpu
    sys.modules['pytorch3d.io'].load_obj = lambda *a, **k: None
    _installed = True


def ref_ctx():
    install()
    return CudaToCpu()
