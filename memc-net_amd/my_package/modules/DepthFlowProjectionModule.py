# modules/DepthFlowProjectionModule.py -- wrapper the reference lacks for its DepthFlowProjectionLayer_* C entry points
from torch.nn import Module
from my_package.functions.DepthFlowProjectionLayer import DepthFlowProjectionLayer


class DepthFlowProjectionModule(Module):
    def __init__(self, requires_grad=True):
        super(DepthFlowProjectionModule, self).__init__()
        self.f = DepthFlowProjectionLayer(requires_grad)

    def forward(self, input1, input2):
        return self.f(input1, input2)
