"""ctypes binding of libd3h_hip.so (C ABI declared in include/d3h.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised
(the reference raises through TORCH_CHECK, render/renderutils/c_src/torch_bindings.cpp:27-31).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libd3h_hip.so')
_lib = None

_I64 = ctypes.c_int64
_I32 = ctypes.c_int
_F32 = ctypes.c_float
_PTR = ctypes.c_void_p


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'd3h: {LIB_PATH} not built -- run __graft_entry__.build() '
                               f'(python d3human-code_amd/d3h/build.py); there is no CPU fallback')
        _lib = ctypes.CDLL(LIB_PATH)
        for name in ('d3h_sdf_mlp_wpack_floats',):
            getattr(_lib, name).restype = _I64
        _lib.d3h_sdf_mlp_act_floats.restype = _I64
        _lib.d3h_sdf_mlp_act_floats.argtypes = [_I64]
    return _lib


def ptr(t):
    """device pointer of a contiguous tensor (None -> NULL)"""
    if t is None:
        return None
    assert t.is_contiguous(), 'd3h: tensor must be contiguous'
    return _PTR(t.data_ptr())


def stream():
    return _PTR(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'd3h: {what} failed with code {rc}' + (' (bad argument)' if rc < 0 else ' (hipError_t)'))


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('d3h: tensors must live on the GPU (no CPU path in the product)')


def i64(v):
    return _I64(int(v))


def i32(v):
    return _I32(int(v))


def f32(v):
    return _F32(float(v))
