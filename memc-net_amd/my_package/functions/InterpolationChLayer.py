"""InterpolationChLayer ("Interpolation_ch") -- plain bilinear backward-warp for any channel count.

The reference ships the C entry points (my_lib_cuda.h:53-68, my_lib.h:49-61) but no Python wrapper; this one
follows the pattern of its InterpolationLayer.py.  On the GPU both share one kernel
(my_lib_cuda.c:519,579).
"""
from .InterpolationLayer import _make_bilinear_function

_InterpolationChFunction = _make_bilinear_function(
    "InterpolationChLayer", "InterpolationChLayer_gpu_forward", "InterpolationChLayer_gpu_backward")


class InterpolationChLayer(object):
    def __init__(self):
        super(InterpolationChLayer, self).__init__()

    def __call__(self, input1, input2):
        return _InterpolationChFunction.apply(input1, input2)

    forward = __call__
