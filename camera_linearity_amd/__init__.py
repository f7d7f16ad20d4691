"""camera_linearity_amd - MI355X-native HDR-merge / linearization engine.

A drop-in for one path of samivout/camera_linearity: ICRF-LUT linearization, Gaussian-weighted HDR
merge, first-order uncertainty propagation, dark-frame hot-pixel filtering and flat-field
correction, behind the reference's Measurand / ImageSet / ExposureSeries API. The arithmetic runs in
hand-written HIP kernels for gfx950 (csrc/), reached through the C ABI of include/hdrmerge.h.
There is no CPU fallback: importing the compute modules without the built library raises.
"""
__version__ = "0.1.0"

from . import settings  # noqa: F401
