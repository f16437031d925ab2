"""FlowUpsample4Layer -- EXTENSION, no reference counterpart.

The prologue of FlowProjection in the networks (networks/MEMC_Net_star.py:172-176):

    F.interpolate(mul * flow / div, scale_factor=4, mode="bilinear", align_corners=align_corners)

as one kernel: the quarter-resolution flow is read, scaled and upsampled, every output element written once
(instead of a scaling kernel, a division kernel and torch's upsampling kernel).  Same sampling rule as torch; results
agree with the torch expression to the last bit or two (fused multiply-adds may contract differently).
Differentiable: the backward pass is ATen's upsample_bilinear2d_backward times mul / div.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import my_package._ext.my_lib as my_lib
from ._common import check, f32c, require_gpu


class _FlowUpsample4Function(Function):
    @staticmethod
    def forward(ctx, flow, mul, div, align_corners):
        require_gpu("FlowUpsample4Layer", flow)
        flow = f32c(flow)
        B, C, h, w = flow.shape
        out = torch.empty((B, C, 4 * h, 4 * w), dtype=flow.dtype, device=flow.device)   # every element is written
        check(my_lib.FlowUpsample4Layer_gpu_forward(flow, out, mul, div, align_corners), "FlowUpsample4Layer_gpu_forward")
        ctx.geom = (tuple(flow.shape), float(mul), float(div), bool(align_corners))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gradoutput):
        shape, mul, div, align = ctx.geom
        g = torch.ops.aten.upsample_bilinear2d_backward(f32c(gradoutput), [4 * shape[2], 4 * shape[3]], list(shape),
                                                        align, 4.0, 4.0)
        return g * (mul / div), None, None, None


class FlowUpsample4Layer(object):
    """`FlowUpsample4Layer(mul, div, align_corners)(flow)`"""

    def __init__(self, mul=1.0, div=1.0, align_corners=False):
        self.mul, self.div, self.align_corners = float(mul), float(div), bool(align_corners)

    def __call__(self, flow):
        return _FlowUpsample4Function.apply(flow, self.mul, self.div, self.align_corners)
