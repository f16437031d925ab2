"""`FilterInterpolationModule()(input1, input2, input3)` -- image [B,C,H,W], flow [B,2,H,W], taps [B,fs*fs,H,W];
the surface of the reference's module of this name."""
from my_package.functions.FilterInterpolationLayer import FilterInterpolationLayer
from ._operator_module import OperatorModule


class FilterInterpolationModule(OperatorModule):
    layer = FilterInterpolationLayer

    def __init__(self):
        OperatorModule.__init__(self)
        self._bind()

    def forward(self, input1, input2, input3):
        return self.f(input1, input2, input3)
