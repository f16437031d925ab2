"""FilterInterpolationCtxBlendLayer -- EXTENSION, no reference counterpart.

`MEMC_Net_star` warps each frame AND its 64-channel context features with the same flow and the same 16 filter
planes, then blends the two warped frames (networks/MEMC_Net_star.py:273-285, :277):

    blended  = occlusion0 * FI(input0, flow0, filter0) + occlusion1 * FI(input2, flow1, filter1)
    context0 = FI(ctx0, flow0, filter0).detach()
    context2 = FI(ctx2, flow1, filter1).detach()

As four launches flow + taps (72 B per site and direction) are streamed twice per direction.  Here one launch per
direction warps the frame and its context together; the second one also applies the blend, with the first one's
warped frame as `prev` (1236 instead of 1356 bytes per site and frame pair; identical results).

Differentiable in the frames, flows, filters and occlusions exactly like FilterInterpolationBlendLayer (the backward
pass goes through the reference-API entry points); the context outputs carry no gradient, as in the reference
(`.detach()`, :284-285).  Shapes the fused kernel does not cover (context channels not a multiple of 4, filters other
than 4x4, widths or storage offsets that are not 16-byte aligned) are composed from the separate layers -- same
values, no error.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu
from .FilterInterpolationBlendLayer import FilterInterpolationBlendLayer, _blend_backward
from .FilterInterpolationLayer import FilterInterpolationLayer


def fused_supported(input0, ctx0, filter0, occlusion0, *others):
    ok = (input0.size(1) == 3 and filter0.size(1) == 16 and input0.size(3) % 4 == 0 and ctx0.size(1) >= 4
          and ctx0.size(1) % 4 == 0 and occlusion0.size(1) == 1)
    return ok and all(t.is_cuda and t.data_ptr() % 16 == 0 for t in (input0, ctx0, filter0, occlusion0) + others)


class _CtxBlendFunction(Function):
    @staticmethod
    def forward(ctx, input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        args = (input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)
        require_gpu("FilterInterpolationCtxBlendLayer", *args)
        input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occ0, occ1 = (f32c(t) for t in args)
        warped0 = torch.empty_like(input0)                      # every element of every output is written
        c0w, c2w, blended = torch.empty_like(ctx0), torch.empty_like(ctx2), torch.empty_like(input0)
        check(my_lib.FilterInterpolationCtxLayer_gpu_forward(input0, ctx0, flow0, filter0, None, None, None,
                                                             warped0, c0w), "FilterInterpolationCtxLayer_gpu_forward")
        check(my_lib.FilterInterpolationCtxLayer_gpu_forward(input2, ctx2, flow1, filter1, warped0, occ0, occ1,
                                                             blended, c2w), "FilterInterpolationCtxLayer_gpu_forward")
        ctx.save_for_backward(input0, input2, flow0, flow1, filter0, filter1, occ0, occ1)
        ctx.mark_non_differentiable(c0w, c2w)
        return blended, c0w, c2w

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput, _g_ctx0, _g_ctx2):
        gx0, gx2, gf0, gf1, gk0, gk1, go0, go1 = _blend_backward(ctx.saved_tensors, f32c(gradoutput))
        return gx0, gx2, None, None, gf0, gf1, gk0, gk1, go0, go1


class FilterInterpolationCtxBlendLayer(object):
    """`FilterInterpolationCtxBlendLayer()(input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0,
    occlusion1)` -> (blended, warped ctx0, warped ctx2); the context outputs are detached."""

    def __call__(self, input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        if fused_supported(input0, ctx0, filter0, occlusion0, input2, ctx2, flow0, flow1, filter1, occlusion1):
            return _CtxBlendFunction.apply(input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0,
                                           occlusion1)
        warp = FilterInterpolationLayer()
        blended = FilterInterpolationBlendLayer()(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)
        return blended, warp(ctx0, flow0, filter0).detach(), warp(ctx2, flow1, filter1).detach()
