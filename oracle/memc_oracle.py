"""numpy front-end of the CPU checker (oracle/memc_oracle.c).

TEST INFRASTRUCTURE ONLY -- parity pinned against the reference's own GPU kernels (see the header of
memc_oracle.c, oracle/ref_gpu.py and DESIGN.md "Oracle").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (memc-net_amd/my_package) never does and has no CPU fallback.

Every function takes float32 numpy arrays in NCHW (any b/c/h strides, unit w stride), allocates the
zero-filled outputs exactly like the reference's Python layer does
(my_package/functions/FilterInterpolationLayer.py:26-29,46-48; FlowProjectionLayer.py:27-29,54) and
raises RuntimeError when the C side returns -1.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmemc_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_S = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile libmemc_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "memc_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libmemc_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.memc_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(lib().memc_oracle_num_threads())


def _prep(a):
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    if a.ndim != 4:
        raise ValueError("expected a 4-D NCHW array")
    if a.strides[3] != 4 and a.shape[3] > 1:
        a = np.ascontiguousarray(a)
    elif 1 in a.shape and not a.flags["C_CONTIGUOUS"]:
        # a dimension of one element carries whatever stride numpy last gave it (a width-1 flow built by a transpose:
        # strides (8, 4, 16, 4)): the C checks compare strides across tensors, so such arrays are normalised
        a = np.ascontiguousarray(a)
    return a


def _ptr(a):
    return a.ctypes.data_as(_F)


def _str(a):
    return (ctypes.c_int64 * 4)(*[s // 4 for s in a.strides])


def _check(err, name):
    if err != 0:
        raise RuntimeError("%s returned %d (shape/stride check failed)" % (name, err))


def _dims(a):
    return [ctypes.c_int(int(v)) for v in a.shape]


# ---------------------------------------------------------------- FilterInterpolation
def filter_interpolation_forward(x, flow, filt):
    x, flow, filt = _prep(x), _prep(flow), _prep(filt)
    if flow.shape != (x.shape[0], 2, x.shape[2], x.shape[3]):
        raise RuntimeError("flow shape mismatch")           # my_lib.c:916-922
    out = np.zeros(x.shape, np.float32)
    B, C, H, W = _dims(x)
    err = lib().memc_oracle_filter_interpolation_forward(
        B, C, H, W, ctypes.c_int(filt.shape[1]), _ptr(x), _str(x), _ptr(flow), _str(flow),
        _ptr(filt), _str(filt), _ptr(out), _str(out))
    _check(err, "filter_interpolation_forward")
    return out


def filter_interpolation_backward(x, flow, filt, gout):
    x, flow, filt = _prep(x), _prep(flow), _prep(filt)
    gout = np.ascontiguousarray(_prep(gout))
    x = np.ascontiguousarray(x)            # gradoutput is indexed with input1's strides (my_lib.c:1191)
    g1 = np.zeros(x.shape, np.float32)
    g2 = np.zeros(flow.shape, np.float32)
    g3 = np.zeros(filt.shape, np.float32)
    flow_c, filt_c = np.ascontiguousarray(flow), np.ascontiguousarray(filt)
    B, C, H, W = _dims(x)
    err = lib().memc_oracle_filter_interpolation_backward(
        B, C, H, W, ctypes.c_int(filt.shape[1]), _ptr(x), _str(x), _ptr(flow_c), _str(flow_c),
        _ptr(filt_c), _str(filt_c), _ptr(gout), _ptr(g1), _ptr(g2), _ptr(g3))
    _check(err, "filter_interpolation_backward")
    return g1, g2, g3


# ---------------------------------------------------------------- Interpolation / InterpolationCh
def _bilinear_forward(name, x, flow):
    x, flow = _prep(x), _prep(flow)
    out = np.zeros(x.shape, np.float32)
    B, C, H, W = _dims(x)
    err = getattr(lib(), name)(B, C, H, W, _ptr(x), _str(x), _ptr(flow), _str(flow),
                               _ptr(out), _str(out))
    _check(err, name)
    return out


def _bilinear_backward(name, x, flow, gout):
    x, flow = np.ascontiguousarray(_prep(x)), np.ascontiguousarray(_prep(flow))
    gout = np.ascontiguousarray(_prep(gout))
    g1 = np.zeros(x.shape, np.float32)
    g2 = np.zeros(flow.shape, np.float32)
    B, C, H, W = _dims(x)
    err = getattr(lib(), name)(B, C, H, W, _ptr(x), _str(x), _ptr(flow), _str(flow),
                               _ptr(gout), _ptr(g1), _ptr(g2))
    _check(err, name)
    return g1, g2


def interpolation_forward(x, flow):
    return _bilinear_forward("memc_oracle_interpolation_forward", x, flow)


def interpolation_backward(x, flow, gout):
    return _bilinear_backward("memc_oracle_interpolation_backward", x, flow, gout)


def interpolation_ch_forward(x, flow):
    return _bilinear_forward("memc_oracle_interpolation_ch_forward", x, flow)


def interpolation_ch_backward(x, flow, gout):
    return _bilinear_backward("memc_oracle_interpolation_ch_backward", x, flow, gout)


# ---------------------------------------------------------------- FlowProjection / DepthFlowProjection
def flow_projection_forward(flow, fillhole=0):
    """Returns (output, count).  fillhole=0 is what the reference's CPU function computes; fillhole=1
    adds the hole-filling pass that exists only in the reference's CUDA file."""
    flow = _prep(flow)
    out = np.zeros(flow.shape, np.float32)
    count = np.zeros((flow.shape[0], 1, flow.shape[2], flow.shape[3]), np.float32)
    B, C, H, W = _dims(flow)
    err = lib().memc_oracle_flow_projection_forward(
        B, C, H, W, _ptr(flow), _str(flow), _ptr(count), _str(count), _ptr(out), _str(out),
        ctypes.c_int(int(fillhole)))
    _check(err, "flow_projection_forward")
    return out, count


def flow_projection_backward(flow, count, gout):
    flow, count = np.ascontiguousarray(_prep(flow)), np.ascontiguousarray(_prep(count))
    gout = np.ascontiguousarray(_prep(gout))
    g1 = np.zeros(flow.shape, np.float32)
    B, C, H, W = _dims(flow)
    err = lib().memc_oracle_flow_projection_backward(
        B, C, H, W, _ptr(flow), _str(flow), _ptr(count), _str(count), _ptr(gout), _ptr(g1))
    _check(err, "flow_projection_backward")
    return g1


def depth_flow_projection_forward(flow, depth, fillhole=0):
    flow, depth = _prep(flow), _prep(depth)
    if depth.shape[1] != 1:
        raise RuntimeError("depth must have one channel")    # my_lib.c:1655
    out = np.zeros(flow.shape, np.float32)
    count = np.zeros((flow.shape[0], 1, flow.shape[2], flow.shape[3]), np.float32)
    B, C, H, W = _dims(flow)
    err = lib().memc_oracle_depth_flow_projection_forward(
        B, C, H, W, _ptr(flow), _str(flow), _ptr(depth), _str(depth), _ptr(count), _str(count),
        _ptr(out), _str(out), ctypes.c_int(int(fillhole)))
    _check(err, "depth_flow_projection_forward")
    return out, count


def depth_flow_projection_backward(flow, depth, count, out, gout):
    flow, depth = np.ascontiguousarray(_prep(flow)), np.ascontiguousarray(_prep(depth))
    count, out = np.ascontiguousarray(_prep(count)), np.ascontiguousarray(_prep(out))
    gout = np.ascontiguousarray(_prep(gout))
    g1 = np.zeros(flow.shape, np.float32)
    g2 = np.zeros(depth.shape, np.float32)
    B, C, H, W = _dims(flow)
    err = lib().memc_oracle_depth_flow_projection_backward(
        B, C, H, W, _ptr(flow), _str(flow), _ptr(depth), _str(depth), _ptr(count), _str(count),
        _ptr(out), _ptr(gout), _ptr(g1), _ptr(g2))
    _check(err, "depth_flow_projection_backward")
    return g1, g2
