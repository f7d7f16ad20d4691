"""Factory mirroring modules/measurand_factory.py:10-56 with the HIP backend in the CuPy slot.

`Measurand(val, std, use_cupy=...)` keeps the reference's signature: the flag that selected the
CuPy (device) backend there selects the HIP backend here. There is no NumPy Measurand in this
package - `use_cupy=False` is rejected instead of silently computing on the host.
"""
from __future__ import annotations

import numpy as np
import torch

from .measurand import HipMeasurand

HIP_AVAILABLE = torch.cuda.is_available()


def Measurand(val=None, std=None, use_cupy=True, backend: str | None = None):
    """Factory function to return the Measurand class of the requested backend (measurand_factory.py:10-14)."""
    if backend is None:
        backend = "hip" if use_cupy else "numpy"
    if backend != "hip":
        raise NotImplementedError(
            "camera_linearity_amd only provides the 'hip' backend; use the reference's NumpyMeasurand for host arrays")
    return HipMeasurand(val, std)


def measurand_to_hip(numpy_val, numpy_std=None) -> HipMeasurand:
    """Counterpart of measurand_to_cupy (measurand_factory.py:17-35): upload host arrays (or a reference
    NumpyMeasurand-like object with .val/.std) into a HipMeasurand. uint8 values stay DNs."""
    if hasattr(numpy_val, "val") and hasattr(numpy_val, "std"):
        numpy_val, numpy_std = numpy_val.val, numpy_val.std
    if isinstance(numpy_val, np.ndarray) and numpy_val.dtype == np.uint8:
        return HipMeasurand.from_dn(numpy_val, numpy_std)
    return HipMeasurand(numpy_val, numpy_std)


def measurand_to_numpy(hip_measurand: HipMeasurand):
    """Counterpart of measurand_to_numpy (measurand_factory.py:38-56): (val, std) as host ndarrays."""
    return hip_measurand.to_numpy()
