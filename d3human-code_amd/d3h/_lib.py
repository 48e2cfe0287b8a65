"""ctypes binding of libd3h_hip.so (C ABI declared in include/d3h.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised
(the reference raises through TORCH_CHECK, render/renderutils/c_src/torch_bindings.cpp:27-31).

`_use_emulator_for_tests(path)` exists only so that tests/test_emul_*.py can drive the very same Python
wrappers against tests/emul/libd3h_emul.so (the host emulation of the kernel sources used to debug kernel
logic in the GPU-less dev container).  Nothing in the package, bench.py or __graft_entry__ calls it.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('D3H_LIB_PATH') or os.path.join(_HERE, 'libd3h_hip.so')     # D3H_LIB_PATH: A/B runs of two builds on one box
_lib = None
_emulated = False
ABI_VERSION = 5          # D3H_ABI_VERSION of include/d3h.h these wrappers were written against (csrc/d3h_common.h)

_I64 = ctypes.c_int64
_I32 = ctypes.c_int
_F32 = ctypes.c_float
_PTR = ctypes.c_void_p

_RESTYPE_I64 = ('d3h_sdf_mlp_wpack_floats', 'd3h_sdf_mlp_wpack3_dwords', 'd3h_sdf_mlp_wpackt3_dwords', 'd3h_sdf_mlp_act_floats', 'd3h_sdf_mlp_wpackt_floats', 'd3h_hashgrid_param_floats',
                'd3h_deform_mlp_wpack_floats', 'd3h_deform_mlp_act_floats', 'd3h_deform_mlp_wpackt_floats', 'd3h_sdf_mlp_bwd_scratch_ints',
                'd3h_deform_mlp_bwd_scratch_ints')


def _configure(l):
    for name in _RESTYPE_I64:
        if hasattr(l, name):
            getattr(l, name).restype = _I64
    l.d3h_sdf_mlp_act_floats.argtypes = [_I64]
    l.d3h_deform_mlp_act_floats.argtypes = [_I64]
    l.d3h_sdf_mlp_bwd_scratch_ints.argtypes = [_I64]
    l.d3h_deform_mlp_bwd_scratch_ints.argtypes = [_I64]
    have = l.d3h_abi_version() if hasattr(l, 'd3h_abi_version') else None
    if have != ABI_VERSION:
        raise RuntimeError(f'd3h: the library reports ABI version {have}, these wrappers need {ABI_VERSION}: a stale build -- '
                           f'rebuild with __graft_entry__.build()')
    return l


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'd3h: {LIB_PATH} not built -- run __graft_entry__.build() '
                               f'(python d3human-code_amd/d3h/build.py); there is no CPU fallback')
        _lib = _configure(ctypes.CDLL(LIB_PATH))
    return _lib


def _use_emulator_for_tests(path):
    """TEST ONLY (see module docstring)."""
    global _lib, _emulated
    _lib = _configure(ctypes.CDLL(path))
    _emulated = True


def emulated():
    return _emulated


_keepalive = []


def ptr(t):
    """device pointer of a contiguous tensor (None -> NULL).  The tensor is kept alive until the matching check() so that
    temporaries such as `ptr(g.contiguous())` cannot be freed (and their memory re-used) before the call is enqueued."""
    if t is None:
        return None
    _keepalive.append(t)
    if not t.is_contiguous():
        raise RuntimeError('d3h: tensor must be contiguous')
    if not _emulated and not t.is_cuda:
        raise RuntimeError('d3h: tensors must live on the GPU (the product has no CPU path)')
    return _PTR(t.data_ptr())


def ptr_any(t):
    """device pointer of a DENSE tensor in whatever memory format it has (contiguous or channels-last): the callee is told the layout"""
    if t is None:
        return None
    _keepalive.append(t)
    if not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError('d3h: tensor must be dense (contiguous or channels-last)')
    if not _emulated and not t.is_cuda:
        raise RuntimeError('d3h: tensors must live on the GPU (the product has no CPU path)')
    return _PTR(t.data_ptr())


def stream():
    """the raw handle of torch's current stream on the current device.  NOT torch.cuda.current_stream(): that goes through
    torch.cuda.is_available() -> an environment lookup and a device-count query of the runtime on EVERY call -- ~120 us each on this
    stack, 34 calls = 4 ms of host time per 7.6 ms training iteration (tools/gpu_cpu_profile.py), which had become the bottleneck of
    the step once the GPU work had shrunk below the host's launch loop."""
    if _emulated:
        return None
    return _PTR(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def cur_stream():
    """torch.cuda.current_stream() without its per-call availability probe (see stream()): passing the device index takes the fast path"""
    return torch.cuda.current_stream(torch._C._cuda_getDevice())


class use_stream:
    """`with torch.cuda.stream(s)` without its `torch.cuda.current_stream(None)` (the slow availability-probing path, see stream()): the
    previous stream comes from cur_stream() and both switches go straight to the binding -- ~5 us instead of ~80 per use"""
    __slots__ = ('s', 'prev')

    def __init__(self, s):
        self.s = s

    def __enter__(self):
        self.prev = cur_stream()
        s = self.s
        torch._C._cuda_setStream(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)
        return s

    def __exit__(self, *exc):
        p = self.prev
        torch._C._cuda_setStream(stream_id=p.stream_id, device_index=p.device_index, device_type=p.device_type)
        return False


def check(rc, what):
    _keepalive.clear()
    if rc != 0:
        raise RuntimeError(f'd3h: {what} failed with code {rc}' + (' (bad argument)' if rc < 0 else ' (hipError_t)'))


def i64(v):
    return _I64(int(v))


def i32(v):
    return _I32(int(v))


def f32(v):
    return _F32(float(v))
