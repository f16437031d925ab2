"""`InterpolationChModule()(input1, input2)` -- bilinear warp, any channel count.  The reference ships the C entry
points (`InterpolationChLayer_*`, my_lib.h:49-61) but no Python wrapper for them; this is that wrapper."""
from ._operator_module import operator_module

InterpolationChModule = operator_module("InterpolationChModule", ("input1", "input2"))
