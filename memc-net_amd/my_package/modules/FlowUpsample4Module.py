"""`FlowUpsample4Module(mul, div, align_corners)(flow)` -- EXTENSION (no reference module of this name): the scaling and
x4 bilinear upsampling in front of FlowProjection (networks/MEMC_Net_star.py:172-176) as one kernel
(functions/FlowUpsample4Layer.py)."""
from ._operator_module import operator_module

FlowUpsample4Module = operator_module("FlowUpsample4Module", ("flow",),
                                      (("mul", 1.0), ("div", 1.0), ("align_corners", False)))
