"""FilterInterpolationBlendLayer -- EXTENSION, no reference counterpart.

    out = occlusion0 * FilterInterpolation(input0, flow0, filter0)
        + occlusion1 * FilterInterpolation(input2, flow1, filter1)

i.e. `FilterInterpolate` of networks/MEMC_Net_star.py:264-277 as ONE kernel (the two warped frames are never
written).  Differentiable: the backward pass goes through the reference-API entry points (two forward
recomputations + two backward launches; nothing but the inputs is kept for it).

The fused kernel covers what the networks use (RGB, 4x4 filters, widths a multiple of 4); any other shape is
composed from FilterInterpolationLayer calls -- same values, no error.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu
from .FilterInterpolationLayer import FilterInterpolationLayer


def fused_supported(input0, filter0, *others):
    """What the fused kernel covers (the C side returns -1 for anything else): RGB, 16 taps, width a multiple of 4,
    a 1-channel occlusion, 16-byte aligned base pointers.  `others`: the remaining tensors of the call (their
    alignment and, for [B, 1, H, W] occlusions, their channel count are checked when given)."""
    if not (input0.size(1) == 3 and filter0.size(1) == 16 and input0.size(3) % 4 == 0):
        return False
    for t in (input0, filter0) + others:
        if t.is_cuda and t.data_ptr() % 16 != 0:            # e.g. a contiguous view with an odd storage offset
            return False
    return all(o.size(1) == 1 for o in others[-2:]) if len(others) >= 2 else True


def _warp(x, flow, filt):
    out = torch.empty_like(x)                                # every element is written
    check(my_lib.FilterInterpolationLayer_gpu_forward(x, flow, filt, out), "FilterInterpolationLayer_gpu_forward")
    return out


def _blend_backward(saved, gradoutput):
    """gradients of occ0 * FI(in0, flow0, filt0) + occ1 * FI(in2, flow1, filt1) w.r.t. its eight inputs, through the
    reference-API entry points (two forward recomputations + two backward launches)"""
    input0, input2, flow0, flow1, filter0, filter1, occ0, occ1 = saved
    grads = []
    for x, flow, filt, occ in ((input0, flow0, filter0, occ0), (input2, flow1, filter1, occ1)):
        warped = _warp(x, flow, filt)                        # recomputed, not stored by the forward pass
        g_occ = (gradoutput * warped).sum(dim=1, keepdim=True)
        g_warp = (gradoutput * occ).contiguous()
        g_x, g_flow, g_filt = torch.zeros_like(x), torch.empty_like(flow), torch.empty_like(filt)
        check(my_lib.FilterInterpolationLayer_gpu_backward(x, flow, filt, g_warp, g_x, g_flow, g_filt),
              "FilterInterpolationLayer_gpu_backward")
        grads.append((g_x, g_flow, g_filt, g_occ))
    (gx0, gf0, gk0, go0), (gx2, gf1, gk1, go1) = grads
    return gx0, gx2, gf0, gf1, gk0, gk1, go0, go1


class _FilterInterpolationBlendFunction(Function):
    @staticmethod
    def forward(ctx, input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        args = (input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)
        require_gpu("FilterInterpolationBlendLayer", *args)
        args = tuple(f32c(t) for t in args)
        output = torch.empty_like(args[0])                   # every element is written
        check(my_lib.FilterInterpolationBlendLayer_gpu_forward(*args, output),
              "FilterInterpolationBlendLayer_gpu_forward")
        ctx.save_for_backward(*args)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput):
        return _blend_backward(ctx.saved_tensors, f32c(gradoutput))


class FilterInterpolationBlendLayer(object):
    """`FilterInterpolationBlendLayer()(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)`"""

    def __call__(self, input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        if fused_supported(input0, filter0, input2, flow0, flow1, filter1, occlusion0, occlusion1):
            return _FilterInterpolationBlendFunction.apply(input0, input2, flow0, flow1, filter0, filter1,
                                                           occlusion0, occlusion1)
        warp = FilterInterpolationLayer()
        return occlusion0 * warp(input0, flow0, filter0) + occlusion1 * warp(input2, flow1, filter1)
