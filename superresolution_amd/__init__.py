"""MI355X-native implementation of the convolutional super-resolution hot path.

The package mirrors the reference's module-level interface for this path
(``SuperSampler``, ``RepRFN``, ``backward_warping``, ``make_coord``) on top of
``libsrhip.so``, a C-ABI library of hand-written gfx950 kernels.  There is no
CPU fallback: importing the operators without the built library raises.
"""
from . import _lib  # noqa: F401  (loads libsrhip.so, raises if it is missing)
from . import functional  # noqa: F401
from .model import SuperSampler  # noqa: F401
from .recurrent import RecurrentSuperSampler  # noqa: F401
from .reprfn import RepRFN  # noqa: F401

__all__ = ["functional", "SuperSampler", "RecurrentSuperSampler", "RepRFN"]
