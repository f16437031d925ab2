"""`InterpolationModule()(input1, input2)` -- bilinear warp of an RGB image by a flow field.

    input1  [B, 3, H, W]  image (the reference's C layer rejects any other channel count, my_lib.c:450;
                          InterpolationChModule is the any-channel sibling)
    input2  [B, 2, H, W]  flow, channel 0 = dx, channel 1 = dy
    returns [B, 3, H, W]  out(y, x) = bilinear(image, (x + dx, y + dy)); 0 where that point leaves the image

Surface of the reference's module of this name (used by networks/MEMC_Net_VE.py:454-497)."""
from ._operator_module import operator_module

InterpolationModule = operator_module("InterpolationModule", ("input1", "input2"))
