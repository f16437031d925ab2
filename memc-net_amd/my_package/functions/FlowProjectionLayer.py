"""FlowProjectionLayer -- forward-splat of -flow to the intermediate frame, count-normalised, optionally
hole-filled.

Mirrors my_package/functions/FlowProjectionLayer.py of the reference: `FlowProjectionLayer(requires_grad)`
then `layer(input1)`; `fillhole = 1 if requires_grad == False else 0` (:15), so holes are filled only at
inference; `count` is kept for backward (:37).  Deliberate differences as listed in
FilterInterpolationLayer.py of this directory.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu


class _FlowProjectionFunction(Function):
    @staticmethod
    def forward(ctx, input1, fillhole):
        require_gpu("FlowProjectionLayer", input1)
        input1 = f32c(input1)
        # the reference zero-fills both (:27-28); the forward pass here DEFINES every element of them on every
        # path (tests/test_gpu_parity.py::test_projection_forward_needs_no_zero_fill): no memsets
        count = input1.new_empty((input1.size(0), 1, input1.size(2), input1.size(3)))
        output = torch.empty_like(input1)
        # a workspace from torch's allocator instead of a block the library keeps (include/memc_warp.h, "EXTENSION:
        # workspace"): the call owns nothing afterwards and a HIP-graph capture records the same kernels as an eager call
        ws = my_lib.flow_projection_workspace(input1, fillhole)
        err = my_lib.FlowProjectionLayer_gpu_forward_ws(input1, count, output, int(fillhole), ws)
        check(err, "FlowProjectionLayer_gpu_forward_ws")
        ctx.save_for_backward(input1, count)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput):
        input1, count = ctx.saved_tensors
        gradoutput = f32c(gradoutput)
        gradinput1 = torch.empty_like(input1)     # reference zero-fills (:54); the kernel stores every element
        err = my_lib.FlowProjectionLayer_gpu_backward(input1, count, gradoutput, gradinput1)
        check(err, "FlowProjectionLayer_gpu_backward")
        return gradinput1, None


class FlowProjectionLayer(object):
    def __init__(self, requires_grad):
        super(FlowProjectionLayer, self).__init__()
        self.requires_grad = requires_grad

    def __call__(self, input1):
        self.fillhole = 1 if self.requires_grad == False else 0    # noqa: E712 -- as the reference, :15
        return _FlowProjectionFunction.apply(input1, self.fillhole)

    forward = __call__
