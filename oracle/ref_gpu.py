"""oracle/ref_gpu.py -- TEST INFRASTRUCTURE ONLY: ctypes access to oracle/_ref/libmemc_ref_gpu.so, i.e. the
reference's OWN GPU kernels and launchers (my_package/src/my_lib_kernel.cu, compiled for gfx950 by `make -C oracle
ref`: hipify-perl + hipcc of the image, from the reference tree; nothing of it is kept in this repository).

It is the strongest checker there is for this path -- the north star asks for "outputs that match the reference CUDA
kernels" -- and the only executable form of the hole-filling pass.  Used by tests/test_gpu_reference.py (live, on the
GPU box) and by tests/golden/make_golden_ref_gpu.py (which records its outputs as fixtures so that the CPU oracle can
be pinned without a GPU).  Never imported by the product.

Argument lists are my_lib_kernel.h:67-220 verbatim (int strides; `cudaStream_t` became `hipStream_t`).  The wrappers
take contiguous float32 CUDA tensors and zero-fill outputs / gradients as the reference's Python layer does."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmemc_ref_gpu.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)        # RTLD_LOCAL: its symbol names equal the product library's
    return _lib


def _call(name, ints, tensors_strided, tensors_plain):
    """ints: the leading int arguments after (stream, nElement); tensors_strided: tensors whose four strides are
    passed (in order); pointers = every tensor of both lists in call order given by `tensors_plain`."""
    f = getattr(_load(), name)
    f.restype = ctypes.c_int
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [stream, ctypes.c_int(int(tensors_plain[-1].numel()))] + [ctypes.c_int(int(v)) for v in ints]
    for t in tensors_strided:
        args += [ctypes.c_int(int(s)) for s in t.stride()]
    args += [ctypes.c_void_p(t.data_ptr()) for t in tensors_plain]
    err = f(*args)
    if err != 0:
        raise RuntimeError("%s (reference) returned %d" % (name, err))


def _c(*ts):
    out = []
    for t in ts:
        assert t.is_cuda and t.dtype == torch.float32
        out.append(t.contiguous())
    return out


def _dims(x):
    B, C, H, W = x.shape
    return W, H, C, B


def filter_interpolation_forward(x, flow, filt):
    x, flow, filt = _c(x, flow, filt)
    out = torch.zeros_like(x)
    fs = int(float(filt.shape[1]) ** 0.5)
    _call("FilterInterpolationLayer_gpu_forward_kernel", list(_dims(x)) + [fs], [x, flow, filt], [x, flow, filt, out])
    return out


def filter_interpolation_backward(x, flow, filt, gout):
    x, flow, filt, gout = _c(x, flow, filt, gout)
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(flow), torch.zeros_like(filt)
    fs = int(float(filt.shape[1]) ** 0.5)
    _call("FilterInterpolationLayer_gpu_backward_kernel", list(_dims(x)) + [fs], [x, flow, filt],
          [x, flow, filt, gout, g1, g2, g3])
    return g1, g2, g3


def interpolation_forward(x, flow, ch=False):
    x, flow = _c(x, flow)
    out = torch.zeros_like(x)
    _call("Interpolation%sLayer_gpu_forward_kernel" % ("Ch" if ch else ""), _dims(x), [x, flow], [x, flow, out])
    return out


def interpolation_backward(x, flow, gout, ch=False):
    x, flow, gout = _c(x, flow, gout)
    g1, g2 = torch.zeros_like(x), torch.zeros_like(flow)
    _call("Interpolation%sLayer_gpu_backward_kernel" % ("Ch" if ch else ""), _dims(x), [x, flow], [x, flow, gout, g1, g2])
    return g1, g2


def flow_projection_forward(flow, fillhole):
    (flow,) = _c(flow)
    count = flow.new_zeros((flow.shape[0], 1, flow.shape[2], flow.shape[3]))
    out = torch.zeros_like(flow)
    _call("FlowProjection_gpu_forward_kernel", list(_dims(flow)) + [int(fillhole)], [flow, count], [flow, count, out])
    return out, count


def flow_projection_backward(flow, count, gout):
    flow, count, gout = _c(flow, count, gout)
    g1 = torch.zeros_like(flow)
    _call("FlowProjection_gpu_backward_kernel", _dims(flow), [flow, count], [flow, count, gout, g1])
    return g1


def depth_flow_projection_forward(flow, depth, fillhole):
    flow, depth = _c(flow, depth)
    count = torch.zeros_like(depth)
    out = torch.zeros_like(flow)
    _call("DepthFlowProjection_gpu_forward_kernel", list(_dims(flow)) + [int(fillhole)], [flow, depth, count],
          [flow, depth, count, out])
    return out, count


def depth_flow_projection_backward(flow, depth, count, fwd_out, gout):
    flow, depth, count, fwd_out, gout = _c(flow, depth, count, fwd_out, gout)
    g1, g2 = torch.zeros_like(flow), torch.zeros_like(depth)
    _call("DepthFlowProjection_gpu_backward_kernel", _dims(flow), [flow, depth, count],
          [flow, depth, count, fwd_out, gout, g1, g2])
    return g1, g2
