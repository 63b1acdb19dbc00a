"""minddet_amd -- MI355X-native detection hot path behind minddet's operator/model-build surface.

Python host on PyTorch-ROCm (device memory, streams, torch.distributed only) calling
libminddet_hip.so, a C-ABI library of hand-written HIP kernels for gfx950.  There is no CPU
fallback: importing the ops without the built library raises.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so until first use)
