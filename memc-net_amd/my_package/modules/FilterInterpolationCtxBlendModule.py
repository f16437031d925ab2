"""`FilterInterpolationCtxBlendModule()(input0, input2, ctx0, ctx2, flow0, flow1, filter0, filter1, occlusion0,
occlusion1)` -- EXTENSION (no reference module of this name): both frames AND their context features warped with one
stream of flow + filter taps per direction, plus the occlusion-weighted blend
(functions/FilterInterpolationCtxBlendLayer.py)."""
from ._operator_module import operator_module

FilterInterpolationCtxBlendModule = operator_module(
    "FilterInterpolationCtxBlendModule",
    ("input0", "input2", "ctx0", "ctx2", "flow0", "flow1", "filter0", "filter1", "occlusion0", "occlusion1"))
