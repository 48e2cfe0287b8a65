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
ABI_VERSION = 7          # D3H_ABI_VERSION of include/d3h.h these wrappers were written against (csrc/d3h_common.h)

_I64 = ctypes.c_int64
_I32 = ctypes.c_int
_F32 = ctypes.c_float
_PTR = ctypes.c_void_p

_RESTYPE_I64 = ('d3h_sdf_mlp_wpack_floats', 'd3h_sdf_mlp_wpack3_dwords', 'd3h_sdf_mlp_wpackh2_dwords', 'd3h_sdf_mlp_wpackt3_dwords', 'd3h_sdf_mlp_wpackth2_dwords', 'd3h_sdf_mlp_act_floats', 'd3h_sdf_mlp_wpackt_floats', 'd3h_hashgrid_param_floats',
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


# ---- zero-filled tensors without a fill launch each ---------------------------------------------------------------------------------
# A config-3 step asked torch for ~20 zero-filled tensors (gradient accumulators of the atomic-scatter kernels, padded index lists): ~20 fill
# kernels of ~5 us on the launch-bound parts of the step (profiles/r5_bench_config3_per_iteration_serialised.csv: FillFunctor x 20).  zeros() carves
# them from a slab that ONE fill zeroed: 32 MB per (device, stream) -- the same stream semantics as torch.zeros, whose fill also runs on the
# current stream -- replaced by a fresh one when used up (never re-used: a carved tensor keeps its slab alive, like a view its base).  The carved
# tensors are not views (Tensor.set_ on the slab's storage): each has its own autograd version counter, so an in-place op on one cannot
# invalidate another that an autograd node saved.  D3H_ZERO_SLAB=0: plain torch.zeros.
# (a tensor that escapes -- a gradient adopted as `.grad`, a ctx buffer of a retained graph -- pins its whole slab, and torch.save of such a
# tensor would serialise the slab's storage: checkpoints are written from parameters and optimiser state, never from `.grad`
# (d3h/checkpoint.py).  The slab stays at 32 MB because the step carves pieces of 3.1 MB (d(grid positions)) and 4.3 MB (the texture tables'
# gradient) from it -- a quarter of a smaller slab would send them back to one fill launch each.  The per-stream table now evicts the least
# recently used entry instead of dropping every stream's live slab.)
ZERO_SLAB = os.environ.get('D3H_ZERO_SLAB', '1') != '0'
SLAB_BYTES = 32 << 20
import collections as _collections          # noqa: E402
_slabs = _collections.OrderedDict()
_ITEM = {}
SLAB_STATS = {'carved': 0, 'slabs': 0, 'plain': 0}


def zeros(shape, dtype=torch.float32, device=None):
    if isinstance(shape, int):
        shape = (shape,)
    shape = tuple(int(v) for v in shape)
    dev = torch.device(device) if device is not None else torch.device('cpu')
    n = 1
    for v in shape:
        n *= v
    item = _ITEM.get(dtype)
    if item is None:
        item = _ITEM[dtype] = torch.empty(0, dtype=dtype).element_size()
    nbytes = n * item
    if not ZERO_SLAB or _emulated or dev.type != 'cuda' or nbytes == 0 or nbytes > SLAB_BYTES // 4:
        SLAB_STATS['plain'] += 1
        return torch.zeros(shape, dtype=dtype, device=dev)
    idx = dev.index if dev.index is not None else torch._C._cuda_getDevice()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    slab = _slabs.get(key)
    if slab is not None:
        _slabs.move_to_end(key)
    off = 0 if slab is None else (slab[1] + 255) & ~255
    if slab is None or off + nbytes > SLAB_BYTES:
        if slab is None and len(_slabs) >= 16:         # streams come and go (tests): do not keep a slab per stream handle ever seen
            _slabs.popitem(last=False)
        slab = _slabs[key] = [torch.zeros(SLAB_BYTES, dtype=torch.uint8, device=torch.device('cuda', idx)), 0]
        SLAB_STATS['slabs'] += 1
        off = 0
    t = torch.empty(0, dtype=dtype, device=slab[0].device).set_(slab[0].untyped_storage(), off // item, shape)
    slab[1] = off + nbytes
    SLAB_STATS['carved'] += 1
    return t


def zeros_like(t, dtype=None):
    return zeros(t.shape, dtype or t.dtype, t.device)


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
