"""DepthFlowProjectionLayer -- FlowProjection weighted by a per-pixel depth map.

The reference ships the C entry points (my_lib_cuda.h:101-117, my_lib.h:92-108) but no Python wrapper; this
one follows the pattern of its FlowProjectionLayer.py: `DepthFlowProjectionLayer(requires_grad)` then
`layer(input1, input2)` with input1 = flow [N,2,H,W], input2 = depth [N,1,H,W];
`fillhole = 1 if requires_grad == False else 0`.  Backward needs the forward output (my_lib.c:1844-1869),
which is therefore saved together with `count`.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu


class _DepthFlowProjectionFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, fillhole):
        require_gpu("DepthFlowProjectionLayer", input1, input2)
        input1, input2 = f32c(input1), f32c(input2)
        count = input1.new_empty((input1.size(0), 1, input1.size(2), input1.size(3)))   # defined by the forward pass
        output = torch.empty_like(input1)                                               # (see FlowProjectionLayer)
        # a workspace from torch's allocator instead of a block the library keeps (include/memc_warp.h, "EXTENSION:
        # workspace"): the call owns nothing afterwards and a HIP-graph capture records the same kernels as an eager call
        ws = my_lib.flow_projection_workspace(input1, fillhole, depth=True)
        err = my_lib.DepthFlowProjectionLayer_gpu_forward_ws(input1, input2, count, output, int(fillhole), ws)
        check(err, "DepthFlowProjectionLayer_gpu_forward_ws")
        ctx.save_for_backward(input1, input2, count, output)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput):
        input1, input2, count, output = ctx.saved_tensors
        gradoutput = f32c(gradoutput)
        gradinput1 = torch.empty_like(input1)     # stored by the kernel, element by element (see FlowProjectionLayer)
        gradinput2 = torch.empty_like(input2)
        err = my_lib.DepthFlowProjectionLayer_gpu_backward(
            input1, input2, count, output, gradoutput, gradinput1, gradinput2)
        check(err, "DepthFlowProjectionLayer_gpu_backward")
        return gradinput1, gradinput2, None


class DepthFlowProjectionLayer(object):
    def __init__(self, requires_grad):
        super(DepthFlowProjectionLayer, self).__init__()
        self.requires_grad = requires_grad

    def __call__(self, input1, input2):
        self.fillhole = 1 if self.requires_grad == False else 0    # noqa: E712
        return _DepthFlowProjectionFunction.apply(input1, input2, self.fillhole)

    forward = __call__
