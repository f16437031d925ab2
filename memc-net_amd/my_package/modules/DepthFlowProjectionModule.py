"""`DepthFlowProjectionModule(requires_grad=True)(input1, input2)` -- flow projection weighted by a depth map
[B,1,H,W].  The reference ships the C entry points (`DepthFlowProjectionLayer_*`, my_lib.h:92-108) but no Python
wrapper; this is that wrapper, with FlowProjectionModule's constructor."""
from ._operator_module import operator_module

DepthFlowProjectionModule = operator_module("DepthFlowProjectionModule", ("input1", "input2"), options=(("requires_grad", True),))
