"""`FlowProjectionModule(requires_grad=True)(input1)` -- flow [B,2,H,W] projected to the middle frame; holes are
filled only when no gradient is wanted (the layer's `fillhole = not requires_grad` policy).  Surface of the
reference's module of this name."""
from my_package.functions.FlowProjectionLayer import FlowProjectionLayer
from ._operator_module import OperatorModule


class FlowProjectionModule(OperatorModule):
    layer = FlowProjectionLayer

    def __init__(self, requires_grad=True):
        OperatorModule.__init__(self)
        self._bind(requires_grad)

    def forward(self, input1):
        return self.f(input1)
