"""FilterInterpolationLayer -- flow sample fused with the per-pixel 4x4 (fs x fs) adaptive filter.

Mirrors my_package/functions/FilterInterpolationLayer.py of the reference: same class name, same call
surface (`FilterInterpolationLayer()(input1, input2, input3)`), same zero-filled caller-allocated buffers,
same gradients (gradinput1, gradinput2, gradinput3).  The reference is a legacy instance-style
autograd.Function (rejected by torch >= 1.3); here the instance is a thin callable over a static Function.

Differences, all deliberate:
  * the reference hands the ORIGINAL (possibly non-contiguous) tensors to C while caching contiguous copies
    (:14-16,29), so a non-contiguous input makes C return -1 and a zero tensor comes back silently;
    here the contiguous copies are what the kernel sees;
  * a non-zero return code raises instead of being ignored (:29) / printed (:56-57);
  * CPU tensors raise (the reference's CPU branch dies with NameError).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu


class _FilterInterpolationFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, input3):
        require_gpu("FilterInterpolationLayer", input1, input2, input3)
        input1, input2, input3 = f32c(input1), f32c(input2), f32c(input3)
        # the reference zero-fills (:26); the forward kernels write EVERY element (invalid sites copy the input,
        # tests/test_gpu_parity.py::test_forward_outputs_need_no_zero_fill), so the memset -- a quarter of the
        # call's time at 720p -- is skipped.  The C entry point still accepts zero-filled buffers, of course.
        output = torch.empty_like(input1)
        err = my_lib.FilterInterpolationLayer_gpu_forward(input1, input2, input3, output)
        check(err, "FilterInterpolationLayer_gpu_forward")
        ctx.save_for_backward(input1, input2, input3)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput):
        input1, input2, input3 = ctx.saved_tensors
        gradoutput = f32c(gradoutput)
        # accumulation target: zero-filled (reference :46) -- except where the library says it STORES gradinput1 on
        # every path (four and more channels with the 4x4 filter; include/memc_warp.h: memc_gradinput1_is_stored):
        # the memset would be a fifth of the call's traffic
        stored = my_lib.gradinput1_is_stored(int(input3.size(1) ** 0.5 + 1e-6), input1.size(1))     # fs as my_lib.c:925
        # the warped frames are data in the reference's networks (MEMC_Net_star.py:266-277): when autograd does not ask
        # for gradinput1, the RGB kernel skips its accumulation and this layer its zero fill (a NULL gradinput1,
        # include/memc_warp.h)
        # (decided up front -- the library serves a NULL gradinput1 for three channels, the 4x4 filter and a width that is a
        # multiple of four, include/memc_warp.h; anything else gets a buffer that is thrown away: no failed first call)
        want1 = ctx.needs_input_grad[0]
        null_ok = input1.size(1) == 3 and input3.size(1) == 16 and input1.size(3) % 4 == 0
        if want1 or not null_ok:
            gradinput1 = torch.empty_like(input1) if stored else torch.zeros_like(input1)
        else:
            gradinput1 = None
        # the reference zero-fills these two as well (:47-48); the backward kernels DEFINE every element of them
        # (invalid sites store zero; tests/test_gpu_parity.py::test_backward_defines_flow_and_tap_gradients), so
        # 72 B/site of memsets -- a seventh of the call at 720p -- are skipped
        gradinput2 = torch.empty_like(input2)
        gradinput3 = torch.empty_like(input3)
        err = my_lib.FilterInterpolationLayer_gpu_backward(
            input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3)
        if err != 0 and gradinput1 is None:
            # the library declines a NULL gradinput1 for reasons this layer does not duplicate (a plane beyond 32-bit offsets,
            # say): once more with a buffer that is thrown away -- one branch, taken on failure only (round-5 review)
            err = my_lib.FilterInterpolationLayer_gpu_backward(
                input1, input2, input3, gradoutput, torch.zeros_like(input1), gradinput2, gradinput3)
        check(err, "FilterInterpolationLayer_gpu_backward")
        return (gradinput1 if want1 else None), gradinput2, gradinput3


class FilterInterpolationLayer(object):
    def __init__(self):
        super(FilterInterpolationLayer, self).__init__()

    def __call__(self, input1, input2, input3):
        return _FilterInterpolationFunction.apply(input1, input2, input3)

    forward = __call__
