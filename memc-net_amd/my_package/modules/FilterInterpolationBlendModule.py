# modules/FilterInterpolationBlendModule.py -- EXTENSION (no reference module of this name): the two adaptive warps of
# a frame pair and their occlusion-weighted blend as one operator (functions/FilterInterpolationBlendLayer.py)
from torch.nn import Module
from my_package.functions.FilterInterpolationBlendLayer import FilterInterpolationBlendLayer


class FilterInterpolationBlendModule(Module):
    def __init__(self):
        super(FilterInterpolationBlendModule, self).__init__()
        self.f = FilterInterpolationBlendLayer()

    def forward(self, input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1):
        return self.f(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)
