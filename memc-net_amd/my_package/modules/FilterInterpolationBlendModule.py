"""`FilterInterpolationBlendModule()(input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1)` --
EXTENSION (no reference module of this name): the two adaptive warps of a frame pair and their occlusion-weighted
blend as one operator (functions/FilterInterpolationBlendLayer.py)."""
from ._operator_module import operator_module

FilterInterpolationBlendModule = operator_module("FilterInterpolationBlendModule", ("input0", "input2", "flow0", "flow1", "filter0", "filter1", "occlusion0", "occlusion1"))
