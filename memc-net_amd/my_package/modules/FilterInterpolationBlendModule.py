"""`FilterInterpolationBlendModule()(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)` --
EXTENSION (no reference module of this name): the two adaptive warps of a frame pair and their occlusion-weighted
blend as one operator (functions/FilterInterpolationBlendLayer.py)."""
from my_package.functions.FilterInterpolationBlendLayer import FilterInterpolationBlendLayer
from ._operator_module import OperatorModule


class FilterInterpolationBlendModule(OperatorModule):
    layer = FilterInterpolationBlendLayer

    def __init__(self):
        OperatorModule.__init__(self)
        self._bind()

    def forward(self, input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        return self.f(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)
