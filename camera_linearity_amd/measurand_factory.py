"""Factory mirroring modules/measurand_factory.py:10-56 with the HIP backend in the CuPy slot.

`Measurand(val, std, use_cupy=...)` keeps the reference's signature: the flag that selected the
CuPy (device) backend there selects the HIP backend here; `use_cupy=False` returns the host backend in the reference's
NumpyMeasurand slot (HostMeasurand: NumPy arrays in and out, computed by the host build of the same C ABI). The choice is the
caller's and explicit - the HIP backend never falls back to the host one.
"""
from __future__ import annotations

import numpy as np
import torch

from .measurand import HipMeasurand, HostMeasurand

HIP_AVAILABLE = torch.cuda.is_available()


def Measurand(val=None, std=None, use_cupy=True, backend: str | None = None):
    """Factory function to return the Measurand class of the requested backend (measurand_factory.py:10-14)."""
    if backend is None:
        backend = "hip" if use_cupy else "numpy"
    if backend == "hip":
        return HipMeasurand(val, std)
    if backend == "numpy":
        return HostMeasurand(val, std)
    raise ValueError(f"unknown backend {backend!r} (hip or numpy)")


def measurand_to_hip(numpy_val, numpy_std=None) -> HipMeasurand:
    """Counterpart of measurand_to_cupy (measurand_factory.py:17-35): upload host arrays (or a reference
    NumpyMeasurand-like object with .val/.std) into a HipMeasurand. uint8 values stay DNs."""
    if hasattr(numpy_val, "val") and hasattr(numpy_val, "std"):
        numpy_val, numpy_std = numpy_val.val, numpy_val.std
    if isinstance(numpy_val, np.ndarray) and numpy_val.dtype == np.uint8:
        return HipMeasurand.from_dn(numpy_val, numpy_std)
    return HipMeasurand(numpy_val, numpy_std)


def measurand_to_numpy(hip_measurand: HipMeasurand) -> HostMeasurand:
    """measurand_to_numpy (measurand_factory.py:38-56): a new host Measurand holding copies of the device arrays (8-bit frames stay DNs)."""
    if hip_measurand.backend == "numpy":
        return hip_measurand
    std = None if hip_measurand._std is None else hip_measurand._std.cpu()
    if hip_measurand._dn_t() is not None:
        return HostMeasurand.from_dn(hip_measurand._dn_t().cpu(), std)
    v = hip_measurand._tv()
    return HostMeasurand(None if v is None else v.cpu(), std)


def measurand_to_cupy(numpy_measurand) -> HipMeasurand:
    """measurand_to_cupy (measurand_factory.py:17-35): the device counterpart is the HIP backend."""
    if getattr(numpy_measurand, "backend", None) == "hip":
        return numpy_measurand
    if isinstance(numpy_measurand, HostMeasurand):
        std = None if numpy_measurand._std is None else numpy_measurand._std.to(HipMeasurand._device())
        if numpy_measurand._dn_t() is not None:
            return HipMeasurand.from_dn(numpy_measurand._dn_t().to(HipMeasurand._device()), std)
        v = numpy_measurand._tv()
        return HipMeasurand(None if v is None else v.to(HipMeasurand._device()), std)
    return measurand_to_hip(numpy_measurand)
