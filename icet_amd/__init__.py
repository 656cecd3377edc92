"""icet_amd -- MI355X-native implementation of ICET's per-voxel distribution-matching hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/icet_hip.h),
``api`` (the host-side mirror of the reference's ``ICET`` constructor surface), ``lidar_sim``
(seeded synthetic scans for the BASELINE configs) and ``dist`` (sharding independent pairs over
the GPUs of a node).
"""
from .api import ICET, Context, MultiContext, IcetError, Params, load_library, default_context  # noqa: F401

__all__ = ["ICET", "Context", "MultiContext", "IcetError", "Params", "load_library", "default_context"]
