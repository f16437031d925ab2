"""ctypes loader for libmemc_hip.so -- replaces the reference's cffi loader
(my_package/_ext/my_lib/__init__.py:1-12, which wrapped every symbol of the compiled ``_my_lib`` so that torch
tensors were unwrapped to TH tensor pointers).

Every ``<Op>Layer_gpu_{forward,backward}`` symbol declared in include/memc_warp.h is exposed here under the
reference's name and argument order.  Arguments are torch CUDA float32 tensors (plus the trailing ``fillhole``
int where the reference has one); the return value is the C function's int (0 ok, -1 failed check).  Work is
enqueued on the current HIP stream of the tensors' device, as the reference enqueues on
``THCState_getCurrentStream`` (my_lib_cuda.c:403); nothing synchronises.

The shared library is required: importing this module without it raises ImportError (no fallback).
"""
import ctypes
import os

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libmemc_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libmemc_hip.so not found at %s -- build it with `make -C %s` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'` at the repo root). "
        "There is no CPU or PyTorch fallback for these operators."
        % (LIB_PATH, os.path.join(_PKG_ROOT, "csrc")))

_lib = ctypes.CDLL(LIB_PATH)


class _Tensor4(ctypes.Structure):
    """memc_tensor4 of include/memc_warp.h"""
    _fields_ = [("data", ctypes.c_void_p),
                ("size", ctypes.c_int64 * 4),
                ("stride", ctypes.c_int64 * 4)]


_lib.memc_hip_version.restype = ctypes.c_char_p


def version():
    return _lib.memc_hip_version().decode()


_lib.memc_last_kernel_path.restype = ctypes.c_char_p


def last_kernel_path():
    """The kernel family the most recent operator call of THIS thread took (include/memc_warp.h), e.g. "fi_fwd:tiled_c3";
    families "direct", "generic", "scalar" and "general" are the slow fallbacks (unaligned geometry, fs != 4, no scratch)."""
    return _lib.memc_last_kernel_path().decode()


_lib.memc_gradinput1_is_stored.restype = ctypes.c_int
_lib.memc_gradinput1_is_stored.argtypes = [ctypes.c_int, ctypes.c_int]


def gradinput1_is_stored(filter_size, channel):
    """The library's own answer (include/memc_warp.h): does the backward store gradinput1 (no zero fill needed) or
    accumulate into it?  filter_size 0 = Interpolation / InterpolationCh."""
    return bool(_lib.memc_gradinput1_is_stored(int(filter_size), int(channel)))


def _describe(t, symbol, position):
    """memc_tensor4 of a torch tensor.  (This runs for every tensor of every call: the checks in the order that lets a good
    tensor through fastest, names formatted only when something is wrong, sizes and strides fetched as two tuples.)"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 4:
        name = "%s arg %d" % (symbol, position)
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s: expected a torch.Tensor, got %s" % (name, type(t).__name__))
        if not t.is_cuda:
            raise TypeError("%s: expected a CUDA (HIP) tensor; these operators have no CPU path" % name)
        if t.dtype != torch.float32:
            raise TypeError("%s: expected float32, got %s" % (name, t.dtype))
        raise TypeError("%s: expected a 4-D NCHW tensor, got %d-D" % (name, t.dim()))
    d = _Tensor4()
    d.data = t.data_ptr()
    d.size[:] = t.shape
    d.stride[:] = t.stride()
    return d


def _bind(symbol, n_tensors, trailing_int=False, lib=None, optional=()):
    """optional: positions of tensor arguments that may be None (passed to C as NULL; extension entry points only)."""
    cfunc = getattr(lib if lib is not None else _lib, symbol)
    cfunc.restype = ctypes.c_int
    cfunc.argtypes = ([ctypes.c_void_p] + [ctypes.POINTER(_Tensor4)] * n_tensors
                      + ([ctypes.c_int] if trailing_int else []))
    n_args = n_tensors + (1 if trailing_int else 0)
    byref, current_device, current_stream = ctypes.byref, torch.cuda.current_device, torch.cuda.current_stream

    def call(*args):
        if len(args) != n_args:
            raise TypeError("%s takes %d arguments (%d given)" % (symbol, n_args, len(args)))
        cargs = []
        dev = None
        for i in range(n_tensors):
            t = args[i]
            if t is None and i in optional:
                cargs.append(None)
                continue
            cargs.append(byref(_describe(t, symbol, i)))
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise TypeError("%s: all tensors must live on the same device" % symbol)
        if trailing_int:
            cargs.append(int(args[-1]))
        if dev.index == current_device():          # the usual case: no device switch (a context manager costs ~4 us)
            return int(cfunc(current_stream(dev).cuda_stream, *cargs))
        with torch.cuda.device(dev):
            return int(cfunc(current_stream(dev).cuda_stream, *cargs))

    call.__name__ = symbol
    call.__doc__ = "ctypes binding of %s (include/memc_warp.h)" % symbol
    return call


# name -> (number of tensor arguments, has trailing int); order and names as in my_lib_cuda.h:37-117
_SYMBOLS = {
    "InterpolationLayer_gpu_forward": (3, False),
    "InterpolationLayer_gpu_backward": (5, False),
    "InterpolationChLayer_gpu_forward": (3, False),
    "InterpolationChLayer_gpu_backward": (5, False),
    "FilterInterpolationLayer_gpu_forward": (4, False),
    "FilterInterpolationLayer_gpu_backward": (7, False),
    "FlowProjectionLayer_gpu_forward": (3, True),
    "FlowProjectionLayer_gpu_backward": (4, False),
    "DepthFlowProjectionLayer_gpu_forward": (4, True),
    "DepthFlowProjectionLayer_gpu_backward": (7, False),
}
# extensions without a reference counterpart (include/memc_warp.h, "EXTENSION"): fused dual warp + blend; image +
# context warp of one direction in one pass (image, context, flow, filter, prev | None, occlusion_prev | None,
# occlusion_this | None, image_out, context_out)
_EXTENSIONS = {
    "FilterInterpolationBlendLayer_gpu_forward": (9, False),
    "FilterInterpolationCtxLayer_gpu_forward": (9, False),
}
_OPTIONAL = {"FilterInterpolationCtxLayer_gpu_forward": (4, 5, 6),
             "FilterInterpolationLayer_gpu_backward": (4,)}        # gradinput1 = None: the image gradient is not wanted

__all__ = ["version", "last_kernel_path", "LIB_PATH"]
for _name, (_n, _flag) in list(_SYMBOLS.items()) + list(_EXTENSIONS.items()):
    globals()[_name] = _bind(_name, _n, _flag, optional=_OPTIONAL.get(_name, ()))
    __all__.append(_name)


def _bind_ws(symbol, n_tensors, lib=None):
    """<Op>Layer_gpu_forward_ws(tensors..., fillhole, workspace): the (Depth)FlowProjection forward with a caller-supplied
    workspace (include/memc_warp.h, "EXTENSION: workspace").  `workspace`: a contiguous CUDA tensor of at least
    flow_projection_workspace_bytes(...) bytes, allocated on the current stream (torch.empty: capturable, freed with the call)."""
    cfunc = getattr(lib if lib is not None else _lib, symbol)
    cfunc.restype = ctypes.c_int
    cfunc.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(_Tensor4)] * n_tensors + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    byref, current_device, current_stream = ctypes.byref, torch.cuda.current_device, torch.cuda.current_stream

    def call(*args):
        if len(args) != n_tensors + 2:
            raise TypeError("%s takes %d arguments (%d given)" % (symbol, n_tensors + 2, len(args)))
        ws = args[-1]
        if not (isinstance(ws, torch.Tensor) and ws.is_cuda and ws.is_contiguous()):
            raise TypeError("%s: workspace must be a contiguous CUDA tensor" % symbol)
        dev = args[0].device if isinstance(args[0], torch.Tensor) else None
        cargs = []
        for i in range(n_tensors):
            cargs.append(byref(_describe(args[i], symbol, i)))
            if args[i].device != dev:
                raise TypeError("%s: all tensors must live on the same device" % symbol)
        if ws.device != dev:
            raise TypeError("%s: all tensors must live on the same device" % symbol)
        cargs += [int(args[-2]), ws.data_ptr(), ws.numel() * ws.element_size()]
        if dev.index == current_device():
            return int(cfunc(current_stream(dev).cuda_stream, *cargs))
        with torch.cuda.device(dev):
            return int(cfunc(current_stream(dev).cuda_stream, *cargs))

    call.__name__ = symbol
    return call


_WS_SYMBOLS = {"FlowProjectionLayer_gpu_forward_ws": 3, "DepthFlowProjectionLayer_gpu_forward_ws": 4}
for _name, _n in _WS_SYMBOLS.items():
    globals()[_name] = _bind_ws(_name, _n)
    __all__.append(_name)

_lib.memc_flow_projection_workspace_bytes.restype = ctypes.c_size_t
_lib.memc_flow_projection_workspace_bytes.argtypes = [ctypes.c_int] * 5


def flow_projection_workspace_bytes(w, h, batch, fillhole, depth=False):
    """memc_flow_projection_workspace_bytes (include/memc_warp.h): bytes of workspace a (Depth)FlowProjection forward of
    that shape needs."""
    return int(_lib.memc_flow_projection_workspace_bytes(int(w), int(h), int(batch), int(fillhole), int(bool(depth))))


def flow_projection_workspace(like, fillhole, depth=False):
    """A workspace for a (Depth)FlowProjection forward on the flow tensor `like` [N, 2, H, W]: allocated by torch's caching
    allocator on the current stream (stream-ordered, capturable into a HIP graph, returned to the allocator when the last
    reference goes -- the kernels of the call are already queued on that stream by then)."""
    n = flow_projection_workspace_bytes(like.size(3), like.size(2), like.size(0), fillhole, depth)
    return torch.empty((max(n, 256),), dtype=torch.uint8, device=like.device)


__all__ += ["flow_projection_workspace_bytes", "flow_projection_workspace"]


def _bind_upsample():
    """FlowUpsample4Layer_gpu_forward(input, output, mul, div, align_corners) -- extension (include/memc_warp.h)"""
    cfunc = _lib.FlowUpsample4Layer_gpu_forward
    cfunc.restype = ctypes.c_int
    cfunc.argtypes = [ctypes.c_void_p, ctypes.POINTER(_Tensor4), ctypes.POINTER(_Tensor4), ctypes.c_float,
                      ctypes.c_float, ctypes.c_int]

    def call(input, output, mul, div, align_corners):
        a, b = _describe(input, "FlowUpsample4Layer_gpu_forward", 0), _describe(output, "FlowUpsample4Layer_gpu_forward", 1)
        if input.device != output.device:
            raise TypeError("FlowUpsample4Layer_gpu_forward: all tensors must live on the same device")
        with torch.cuda.device(input.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(input.device).cuda_stream)
            return int(cfunc(stream, ctypes.byref(a), ctypes.byref(b), float(mul), float(div), int(bool(align_corners))))

    call.__name__ = "FlowUpsample4Layer_gpu_forward"
    return call


FlowUpsample4Layer_gpu_forward = _bind_upsample()
__all__.append("FlowUpsample4Layer_gpu_forward")


def _cpu_unavailable(symbol):
    def call(*args):
        raise RuntimeError(
            "%s: no CPU implementation is shipped (the reference's Python CPU branches are themselves "
            "unrunnable, e.g. FilterInterpolationLayer.py:23,32); move the tensors to the GPU" % symbol)
    call.__name__ = symbol
    return call


for _name in list(_SYMBOLS):
    _cpu = _name.replace("_gpu_", "_cpu_")
    globals()[_cpu] = _cpu_unavailable(_cpu)
    __all__.append(_cpu)


def calibration_stream(src, dst, reads_per_write):
    """memc_calibration_stream (include/memc_warp.h, measurement aid): dst[i] = sum of `reads_per_write` streams of src,
    16 bytes per lane, non-temporal, the tiled kernels' XCD walk.  src: flat float32 CUDA tensor of reads_per_write *
    dst.numel() elements; dst.numel() a multiple of 4.  Returns the C function's int."""
    if not (src.is_cuda and dst.is_cuda and src.dtype == torch.float32 and dst.dtype == torch.float32):
        raise TypeError("calibration_stream: float32 CUDA tensors")
    if dst.numel() % 4 or src.numel() != int(reads_per_write) * dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise TypeError("calibration_stream: src must hold reads_per_write * dst.numel() contiguous elements")
    cfunc = _lib.memc_calibration_stream
    cfunc.restype = ctypes.c_int
    cfunc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    with torch.cuda.device(src.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
        return int(cfunc(stream, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), dst.numel() // 4,
                         int(reads_per_write)))


__all__.append("calibration_stream")
