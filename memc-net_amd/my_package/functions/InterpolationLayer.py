"""InterpolationLayer -- plain bilinear backward-warp, 3 channels only (my_lib_cuda.c:373).

Mirrors my_package/functions/InterpolationLayer.py of the reference (same class name and call surface:
`InterpolationLayer()(input1, input2)`, gradients for the image and the flow); see
FilterInterpolationLayer.py in this directory for the list of deliberate differences.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu


def _make_bilinear_function(label, fwd_name, bwd_name):
    fwd, bwd = getattr(my_lib, fwd_name), getattr(my_lib, bwd_name)

    class _BilinearFunction(Function):
        @staticmethod
        def forward(ctx, input1, input2):
            require_gpu(label, input1, input2)
            input1, input2 = f32c(input1), f32c(input2)
            output = torch.empty_like(input1)     # reference zero-fills (:25); every element is written (invalid: 0)
            check(fwd(input1, input2, output), fwd_name)
            ctx.save_for_backward(input1, input2)
            return output

        @staticmethod
        @once_differentiable
        def backward(ctx, gradoutput):
            input1, input2 = ctx.saved_tensors
            gradoutput = f32c(gradoutput)
            # accumulation target, zero-filled (reference :40-41) -- except for four and more channels,
            # for which the library STORES it on every path (include/memc_warp.h)
            stored = my_lib.gradinput1_is_stored(0, input1.size(1))     # the library's own rule (include/memc_warp.h)
            gradinput1 = torch.empty_like(input1) if stored else torch.zeros_like(input1)
            gradinput2 = torch.empty_like(input2)               # assigned at every site (invalid: 0)
            check(bwd(input1, input2, gradoutput, gradinput1, gradinput2), bwd_name)
            return gradinput1, gradinput2

    _BilinearFunction.__name__ = "_%sFunction" % label
    return _BilinearFunction


_InterpolationFunction = _make_bilinear_function(
    "InterpolationLayer", "InterpolationLayer_gpu_forward", "InterpolationLayer_gpu_backward")


class InterpolationLayer(object):
    def __init__(self):
        super(InterpolationLayer, self).__init__()

    def __call__(self, input1, input2):
        return _InterpolationFunction.apply(input1, input2)

    forward = __call__
