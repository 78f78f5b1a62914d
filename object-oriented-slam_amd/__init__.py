"""MI355X-native backend for EllipsoidSLAM's hot path (per-frame ellipsoid fit + quadric graph
optimisation).  The compute lives in csrc/ (hand-written HIP for gfx950 behind the C-ABI of
include/esl.h); this package is the thin host-side mirror of the reference's class surface.

The directory name carries a hyphen, so import it with
    importlib.import_module("object-oriented-slam_amd")
"""
from . import abi, lib, synth  # noqa: F401
from .abi import Graph, default_lm_params  # noqa: F401
from .lib import Context, EslError  # noqa: F401

__all__ = ["abi", "lib", "synth", "Graph", "default_lm_params", "Context", "EslError"]
