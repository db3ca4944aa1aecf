"""rlao_amd -- MI355X-native batched adaptive-optics RL environment (drop-in for drl4ao's OOPAO gym env).

The per-step physics runs in hand-written HIP kernels (rlao_amd/csrc -> libaoenv.so, C ABI in
include/aoenv.h); this package holds the Python host: parameter handling, one-off NumPy constants,
the GPU calibration driver and the reference's ``reset_soft() / step()`` surface.
"""
from .calib import AOParams, params_from_args  # noqa: F401

__all__ = ["AOParams", "params_from_args", "BatchedAOEnv", "OOPAO", "TorchWrapper", "TimeDelayEnv"]


def __getattr__(name):
    if name in ("BatchedAOEnv", "OOPAO"):
        from . import env
        return getattr(env, name)
    if name in ("TorchWrapper", "TimeDelayEnv"):
        from . import wrappers
        return getattr(wrappers, name)
    raise AttributeError(name)
