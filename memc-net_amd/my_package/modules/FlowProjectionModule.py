"""`FlowProjectionModule(requires_grad=True)(input1)` -- flow [B,2,H,W] projected to the middle frame; holes are
filled only when no gradient is wanted (the layer's `fillhole = not requires_grad` policy).  Surface of the
reference's module of this name."""
from ._operator_module import operator_module

FlowProjectionModule = operator_module("FlowProjectionModule", ("input1",), options=(("requires_grad", True),))
