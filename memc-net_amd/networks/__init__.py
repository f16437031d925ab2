"""Build-owned network definitions that sit on top of the hot path.

`MEMC_Net_star` and `MEMC_Net`: same constructor, forward signature, return structure and state-dict keys as the
reference's `networks.MEMC_Net_star` (so its checkpoints load), written from scratch on stock `torch.nn` layers
plus this repository's `my_package` operators.  Needed because the reference's files cannot travel to the GPU
box; validated against the reference network itself in `tests/test_network_star.py` (CPU, this container).
"""
from .MEMC_Net import MEMC_Net
from .MEMC_Net_star import MEMC_Net_star
from .inference import interpolate_pairs, pad_amounts
from .replicate import broadcast_module_state, shard_pairs
from .png_io import interpolate_png_tree, read_png, write_png
from .yuv_io import Yuv420Reader, Yuv420Writer, interpolate_yuv_sequence

__all__ = ("MEMC_Net", "MEMC_Net_star", "broadcast_module_state", "shard_pairs", "interpolate_pairs", "pad_amounts",
           "Yuv420Reader", "Yuv420Writer", "interpolate_yuv_sequence", "read_png", "write_png", "interpolate_png_tree")
