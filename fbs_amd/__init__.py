"""fbs_amd -- MI355X-native engine for the particle-Gibbs / CSMC / pMCMC hot path of zgbkdlm/fbs.

``fbs_amd.samplers`` and ``fbs_amd.sdes`` keep the reference's function names and argument orders
on torch (ROCm) tensors and ``uint32[2]`` threefry keys; the device work is hand-written HIP for
gfx950 behind the C ABI of ``include/fbsmi.h`` (``fbs_amd/lib/libfbsmi.so``).  There is no CPU
fallback: without the HIP extension or a GPU every operation raises.
"""
from . import _lib  # noqa: F401
from .ops import PRNGKey, split  # noqa: F401
from . import ops, sdes, samplers  # noqa: F401
from .linear_gaussian import LinearGaussianBridge  # noqa: F401

__all__ = ["ops", "sdes", "samplers", "LinearGaussianBridge", "PRNGKey", "split", "build"]


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc) and return its path."""
    return _lib.build(force=force)
