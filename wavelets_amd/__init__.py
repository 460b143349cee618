"""wavelets_amd - MI355X-native a-trous wavelet engine with the ``watroo`` Python API.

    from wavelets_amd import AtrousTransform, B3spline, Triangle, Coefficients, denoise, wow

mirrors ``from watroo import ...`` (/root/reference/watroo/__init__.py:1-2).  The arithmetic
runs in hand-written HIP kernels (wavelets_amd/csrc) behind the C ABI of
include/watroo_hip.h; see DESIGN.md and INTEGRATION.md.
"""
from .wavelets import *  # noqa: F401,F403
from .wavelets import atrous_convolution, sdev_loc, AbstractScalingFunction  # noqa: F401
from .utils import *  # noqa: F401,F403
from .sequence import map_frames, denoise_many, wow_many, transform_many  # noqa: F401  (sequences of frames: double-buffered over PCIe)

__version__ = '0.1.0'
