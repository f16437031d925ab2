# modules/FilterInterpolationModule.py -- same surface as the reference's module of this name
from torch.nn import Module
from my_package.functions.FilterInterpolationLayer import FilterInterpolationLayer


class FilterInterpolationModule(Module):
    def __init__(self):
        super(FilterInterpolationModule, self).__init__()
        self.f = FilterInterpolationLayer()

    def forward(self, input1, input2, input3):
        return self.f(input1, input2, input3)
