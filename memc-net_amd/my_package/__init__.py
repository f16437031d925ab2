"""Drop-in replacement for the reference's ``my_package`` (MEMC-Net's custom warp / projection operators),
backed by hand-written HIP kernels for MI355X (gfx950) in ``../lib/libmemc_hip.so``.

Put the directory that contains this package (``memc-net_amd/``) on ``sys.path`` -- exactly where the
reference keeps its own ``my_package`` relative to ``networks/`` -- and ``networks/MEMC_Net*.py`` import it
unmodified:

    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    from my_package.modules.FlowProjectionModule import FlowProjectionModule
    from my_package.modules.InterpolationModule import InterpolationModule

Added next to those (the reference has the C entry points, my_lib.h:49-61,92-108, but ships no Python
wrapper): ``DepthFlowProjectionModule`` and ``InterpolationChModule``.

There is no CPU path: the reference's own CPU branches are unrunnable (they never allocate ``output``,
FilterInterpolationLayer.py:23,32) and this package does not add one.  CPU tensors raise.
"""
