"""supervised_dispnet_amd -- MI355X (gfx950) implementation of the zenithfang/supervised_dispnet training hot path.

Host side mirrors the reference's Python interface (models.*, loss_functions.*, inverse_warp.*, layers.*);
everything numerically heavy is a hand-written HIP kernel in libdispnet_hip.so behind a flat C ABI
(include/dispnet_hip.h).  There is no CPU fallback: importing the compute modules without the built library raises.
"""
__version__ = "0.1.0"
