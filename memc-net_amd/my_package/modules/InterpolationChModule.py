"""`InterpolationChModule()(input1, input2)` -- bilinear warp, any channel count.  The reference ships the C entry
points (`InterpolationChLayer_*`, my_lib.h:49-61) but no Python wrapper for them; this is that wrapper."""
from my_package.functions.InterpolationChLayer import InterpolationChLayer
from ._operator_module import OperatorModule


class InterpolationChModule(OperatorModule):
    layer = InterpolationChLayer

    def __init__(self):
        OperatorModule.__init__(self)
        self._bind()

    def forward(self, input1, input2):
        return self.f(input1, input2)
