"""`FilterInterpolationModule()(input1, input2, input3)` -- image [B,C,H,W], flow [B,2,H,W], taps [B,fs*fs,H,W];
the surface of the reference's module of this name."""
from ._operator_module import operator_module

FilterInterpolationModule = operator_module("FilterInterpolationModule", ("input1", "input2", "input3"))
