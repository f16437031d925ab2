"""`DepthFlowProjectionModule(requires_grad=True)(input1, input2)` -- flow projection weighted by a depth map
[B,1,H,W].  The reference ships the C entry points (`DepthFlowProjectionLayer_*`, my_lib.h:92-108) but no Python
wrapper; this is that wrapper, with FlowProjectionModule's constructor."""
from my_package.functions.DepthFlowProjectionLayer import DepthFlowProjectionLayer
from ._operator_module import OperatorModule


class DepthFlowProjectionModule(OperatorModule):
    layer = DepthFlowProjectionLayer

    def __init__(self, requires_grad=True):
        OperatorModule.__init__(self)
        self._bind(requires_grad)

    def forward(self, input1, input2):
        return self.f(input1, input2)
