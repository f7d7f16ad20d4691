"""HipMeasurand - the MI355X backend of the reference's Measurand protocol.

Mirror of AbstractMeasurand / NumpyMeasurand / CupyMeasurand (modules/measurand.py:26-761,
modules/cupy_measurand.py:28-137): a value array and an uncertainty array of the same shape, with
operators that propagate uncertainty to first order. Same method names, argument meaning and error
behaviour; the arrays are torch tensors resident in HBM and every hot-path method launches a
hand-written HIP kernel through the C ABI (engine.py -> libhdrmerge.so). There is no CPU fallback.

Differences from the reference, all from SURVEY.md 3.4 (the reference's multi-channel path does not
run as written):
  A/B  linearize on (..., C) data returns (..., C): out[..., c] = ICRF[idx[..., c], c]; `channels` is
       arange(shape[-1]) and is set by the constructor too.
  F    filter_larger_than_by_map replaces the masked pixels by the k x k median ('reflect').
  H    normalize_by_map uses the integer ROI.
Extra: `val` may be a uint8 DN tensor (the reference accepts integer arrays in linearize,
measurand.py:505); `HipMeasurand.from_dn()` keeps an 8-bit frame as DNs in HBM (1 byte/element
instead of 8) and materialises DN/255 (modules/image_set.py:223) only when `.val` is read.

Rows "next" of SURVEY.md 8f-1: apply_thresholds, compute_difference, interpolate (equal shapes or broadcasting operands),
compute_dimension_statistics (any axis or axis tuple), extract and compute_channel_histogram run as HIP kernels
(csrc/hm_stats.hip, hm_ops.hip); what a kernel does not cover raises (more than 32 channels in apply_thresholds, more
than 4 channels in the histogram, kernel density estimates) - nothing is computed with torch or NumPy arithmetic.
"""
from __future__ import annotations

import math
from typing import List, Optional, Union

import numpy as np
import torch

from . import settings as gs

ScalarType = Union[int, float]
_F64 = torch.float64
_U8 = torch.uint8


def is_broadcastable(shape1, shape2) -> bool:
    """modules/general_functions.py:14-24."""
    if not shape1 or not shape2:
        raise ValueError("Shapes cannot be empty")
    for a, b in zip(tuple(shape1)[::-1], tuple(shape2)[::-1]):
        if not (a == 1 or b == 1 or a == b):
            return False
    return True


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _engine():
    from . import engine          # imports _native: raises ImportError when libhdrmerge.so is missing
    return engine


class AbstractMeasurand:
    """Backend protocol (modules/measurand.py:26-32): subclasses define lib, backend, ArrayType, InputType."""
    lib = None
    backend = None
    fn_median_filter = None
    ArrayType = None
    InputType = None


class HipMeasurand(AbstractMeasurand):
    lib = torch
    backend = "hip"
    ArrayType = torch.Tensor
    InputType = (int, float, torch.Tensor, np.ndarray)

    # ---- backend hooks (HostMeasurand overrides them): where arrays live, how they are handed out, which library computes
    @classmethod
    def _device(cls) -> torch.device:
        return default_device()

    @staticmethod
    def _export(t):
        """What the public properties hand out for an internal tensor: the tensor itself on the HIP backend."""
        return t

    def _eng(self):
        return _engine()

    def _need_resident(self, t: torch.Tensor, what: str) -> None:
        if not t.is_cuda:
            raise RuntimeError(f"{what} needs the image on the device (there is no CPU fallback)")

    # ---------------------------------------------------------------- construction
    def __init__(self, val=None, std=None):
        """modules/measurand.py:695-714: scalars become shape-(1,) float64 arrays; shapes must match.
        NumPy arrays are uploaded silently, as CupyMeasurand does (modules/cupy_measurand.py:66-73)."""
        if val is not None and not isinstance(val, self.InputType) or isinstance(val, bool):
            raise TypeError("Invalid value type.")
        if std is not None and not isinstance(std, self.InputType) or isinstance(std, bool):
            raise TypeError("Invalid std type")
        self._dn = None
        self._val = self._to_tensor(val)
        self._std = self._to_tensor(std, like=self._val)
        if self._val is not None and self._std is not None and self._val.shape != self._std.shape:
            raise ValueError("Value and std shapes must match.")
        self._channels = None
        self._update_channels()
        self._initialized = True

    @classmethod
    def from_dn(cls, dn: torch.Tensor, std=None) -> "HipMeasurand":
        """An 8-bit frame kept as uint8 DNs in HBM; `.val` is DN / MAX_DN on demand (image_set.py:223)."""
        if isinstance(dn, np.ndarray):
            dn = torch.as_tensor(np.ascontiguousarray(dn), device=cls._device())
        if not isinstance(dn, torch.Tensor) or dn.dtype != _U8:
            raise TypeError("from_dn expects a uint8 array")
        m = cls(None, None)
        m._dn = dn if dn.device == cls._device() or cls.backend == "hip" else dn.to(cls._device())
        m._std = m._to_tensor(std, like=m._dn)
        if m._std is not None and m._std.shape != dn.shape:
            raise ValueError("Value and std shapes must match.")
        m._update_channels()
        return m

    @classmethod
    def _to_tensor(cls, x, like: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        if x is None:
            return None
        dev = like.device if like is not None else cls._device()
        if isinstance(x, (int, float)):
            return torch.tensor([x], dtype=_F64, device=dev)
        if isinstance(x, np.ndarray):
            return torch.as_tensor(np.ascontiguousarray(x), device=dev)
        return x

    def _update_channels(self):
        ref = self._val if self._val is not None else self._dn
        if ref is None or ref.dim() == 0:
            self._channels = None
        else:
            self._channels = torch.arange(ref.shape[-1], device=ref.device)       # deviation B

    # ---------------------------------------------------------------- properties (measurand.py:49-84)
    def _tv(self) -> Optional[torch.Tensor]:
        """The value tensor (DN / MAX_DN is materialised on first use for 8-bit frames kept as DNs)."""
        if self._val is None and self._dn is not None:
            self._val = self._eng().u8_to_unit(self._dn)
        return self._val

    def _dn_t(self) -> Optional[torch.Tensor]:
        """uint8 DN tensor when the frame is 8-bit sourced (from_dn, or a uint8 `val`), else None."""
        if self._dn is not None:
            return self._dn
        if self._val is not None and self._val.dtype == _U8:
            return self._val
        return None

    @property
    def val(self):
        return self._export(self._tv())

    @val.setter
    def val(self, value):
        if value is not None and not isinstance(value, (torch.Tensor, np.ndarray)):
            raise TypeError(f"val must be an array or None, got {type(value)} instead.")
        self._dn = None
        self._val = self._to_tensor(value, like=self._std)
        self._update_channels()

    @property
    def dn(self):
        """uint8 DNs when the frame is 8-bit sourced (from_dn, or a uint8 `val`), else None."""
        return self._export(self._dn_t())

    @property
    def std(self):
        return self._export(self._std)

    @std.setter
    def std(self, value):
        if value is not None and not isinstance(value, (torch.Tensor, np.ndarray)):
            raise TypeError(f"std must be an array or None, got {type(value)} instead.")
        ref = self._val if self._val is not None else self._dn
        self._std = self._to_tensor(value, like=ref)

    @property
    def channels(self):
        return self._export(self._channels)

    @channels.setter
    def channels(self, new_channels):
        raise AttributeError("Channels is a read-only attribute, based on the shape of val array.")

    @property
    def shape(self):
        ref = self._val if self._val is not None else self._dn
        return None if ref is None else tuple(ref.shape)

    def __repr__(self):
        value_shape = self.shape if self.shape is not None else "None"
        std_shape = tuple(self._std.shape) if self._std is not None else "None"
        return f"Measurand(backend={self.backend}, value.shape= {value_shape}, std.shape= {std_shape})"

    def __copy__(self):
        m = self.__class__(self._val, self._std)
        m._dn = self._dn
        m._update_channels()
        return m

    def __deepcopy__(self, memo):
        m = self.__class__(None if self._val is None else self._val.clone(), None if self._std is None else self._std.clone())
        m._dn = None if self._dn is None else self._dn.clone()
        m._update_channels()
        memo[id(self)] = m
        return m

    def to_numpy(self):
        """(val, std) as NumPy arrays on the host (D2H copy)."""
        v = self._tv()
        return (None if v is None else v.cpu().numpy()), (None if self._std is None else self._std.cpu().numpy())

    # ---------------------------------------------------------------- operators (measurand.py:106-302)
    def _normalize_input(self, other):
        """modules/measurand.py:281-302."""
        if isinstance(other, AbstractMeasurand):
            if other.backend != self.backend:                      # the reference's backends are sibling classes: isinstance fails there too
                raise TypeError("Invalid other type.")
            normalized_other = other
        elif isinstance(other, self.InputType) and not isinstance(other, bool):
            like = self._val if self._val is not None else self._dn
            normalized_other = self.__class__(self._to_tensor(other, like=like))
        else:
            raise TypeError("Invalid other type.")
        use_std = self._std is not None or normalized_other._std is not None
        return normalized_other, use_std

    def _f64(self):
        v = self._tv()
        return v if v.dtype == _F64 else v.to(_F64)

    def _binary(self, other, op):
        from ._native import HM_OP_ADD  # noqa: F401  (fail early and loudly without the library)
        normalized_other, use_std = self._normalize_input(other)
        x1, x2 = self._f64(), normalized_other._f64()
        if not is_broadcastable(x1.shape, x2.shape):
            raise ValueError("Measurands are not broadcastable.")
        s1 = self._std if use_std else None
        s2 = normalized_other._std if use_std else None
        if x2.device != x1.device:
            x2 = x2.to(x1.device)
            s2 = None if s2 is None else s2.to(x1.device)
        val, std = self._eng().elementwise_binary(op, x1, s1, x2, s2)
        return self.__class__(val, std)

    def __add__(self, other):
        from ._native import HM_OP_ADD
        return self._binary(other, HM_OP_ADD)

    def __sub__(self, other):
        from ._native import HM_OP_SUB
        return self._binary(other, HM_OP_SUB)

    def __mul__(self, other):
        from ._native import HM_OP_MUL
        return self._binary(other, HM_OP_MUL)

    def __rmul__(self, other):
        """modules/measurand.py:213-215: `self * Measurand(other)`."""
        return self * self.__class__(self._to_tensor(other, like=self._val if self._val is not None else self._dn))

    def __truediv__(self, other):
        from ._native import HM_OP_DIV
        return self._binary(other, HM_OP_DIV)

    def __pow__(self, other):
        from ._native import HM_OP_POW
        if isinstance(other, (int, float)) and not isinstance(other, bool):
            # plain scalar exponent (S ** 2, ** (1/2) of the merge loop): the propagation formula of measurand.py:236-239 with
            # s2 = 0, value and derivative without pow() where the exponent allows (hm_pow_scalar)
            val, std = self._eng().pow_scalar(self._f64(), self._std, float(other))
            return self.__class__(val, std)
        return self._binary(other, HM_OP_POW)

    def _unary(self, op):
        val, std = self._eng().elementwise_unary(op, self._f64(), self._std)
        return self.__class__(val, std)

    def __neg__(self):
        from ._native import HM_UOP_NEG
        return self._unary(HM_UOP_NEG)

    def log_e(self):
        from ._native import HM_UOP_LOG_E
        return self._unary(HM_UOP_LOG_E)

    def log_10(self):
        from ._native import HM_UOP_LOG_10
        return self._unary(HM_UOP_LOG_10)

    def zeros_like_measurand(self):
        """modules/measurand.py:304-316."""
        ref = self._val if self._val is not None else self._dn
        new_val = None if ref is None else torch.zeros(ref.shape, dtype=_F64, device=ref.device)
        new_std = None if self._std is None else torch.zeros_like(self._std)
        return self.__class__(new_val, new_std)

    # ---------------------------------------------------------------- hot path
    def linearize(self, ICRF, ICRF_diff=None):
        """modules/measurand.py:471-541 -> new Measurand with the ICRF-mapped values (and ICRF_diff * std)."""
        src = self._dn_t() if self._dn_t() is not None else self._tv()
        val, std = self._eng().linearize(src, self._std, ICRF, ICRF_diff)
        return self.__class__(val, std)

    def lut_index(self):
        """The uint8 LUT index linearize uses (measurand.py:503/505) - exposed for the bit-exact check."""
        if self._dn_t() is not None:
            return self._export(self._dn_t().clone())
        ident = np.zeros((gs.BITS,), dtype=np.float64)
        return self._export(self._eng().linearize(self._tv(), None, ident, return_index=True)[2])

    def apply_gaussian_weight(self):
        """modules/measurand.py:606-618 -> (y, dydx) arrays."""
        src = self._dn_t() if self._dn_t() is not None else self._tv()
        w, dw = self._eng().gaussian_weight(src)
        return self._export(w), self._export(dw)

    def filter_larger_than_by_map(self, map: "HipMeasurand", threshold_value: float):
        """modules/measurand.py:543-557 (intended semantics, deviation F)."""
        eng = self._eng()
        k = gs.MEDIAN_FILTER_KERNEL_SIZE
        dmap = map._dn_t() if map._dn_t() is not None else map._tv()
        src = self._dn_t() if self._dn_t() is not None else self._tv()
        new_val = eng.hot_pixel_filter(src, dmap, threshold_value, k)
        new_std = None if self._std is None else eng.hot_pixel_filter(self._std, dmap, threshold_value, k)
        if new_val.dtype == _U8:
            return self.__class__.from_dn(new_val, new_std)
        return self.__class__(new_val, new_std)

    def normalize_by_map(self, map: "HipMeasurand"):
        """modules/measurand.py:559-604 (integer ROI, deviation H). ROI size from gs.IM_SIZE_X/Y or the map."""
        eng = self._eng()
        fval = map._dn_t() if map._dn_t() is not None else map._tv()
        size_x = gs.IM_SIZE_X or fval.shape[0]
        size_y = gs.IM_SIZE_Y or fval.shape[1]
        x0, x1, y0, y1 = eng.flat_roi_bounds(size_x, size_y, gs.FF_MID_PERCENTAGE)
        means = eng.roi_mean(fval, x0, x1, y0, y1).cpu().numpy()
        std_means = None
        if self._std is not None:
            if map._std is None:
                raise ValueError("flat field needs a std image to propagate uncertainty")
            std_means = eng.roi_mean(map._std, x0, x1, y0, y1).cpu().numpy()
        val, std = eng.normalize_by_map(self._f64(), self._std, fval, map._std, means, std_means)
        return self.__class__(val, std)

    # ---------------------------------------------------------------- "next" rows: every branch below is one HIP kernel of
    # libhdrmerge or an explicit error - there are no torch arithmetic fallbacks (tests/test_gpu_api.py asserts which symbol ran)
    def extract(self, dims=None, axis: Optional[int] = None):
        """modules/measurand.py:352-373: lib.take(val, dims, axis); axis=None indexes the flattened array (hm_take_axis)."""
        if dims is None:
            raise TypeError("extract() needs the indices to take (dims); the reference's lib.take(val, None, axis) raises too")
        target = [dims] if type(dims) is int else list(dims)
        value, std = self._eng().take_axis(self._f64(), self._std, target, axis)
        return self.__class__(value, std)

    def apply_thresholds(self, lower: Optional[List] = None, upper: Optional[List] = None):
        """modules/measurand.py:375-428 (in place)."""
        value = self._f64()
        n = value.shape[-1]
        lower = [None] * n if lower is None else lower
        upper = [None] * n if upper is None else upper
        if len(lower) != n or len(upper) != n:
            raise ValueError("The length of 'lower' and 'upper' must match the size of the independent axis.")
        lo_l = [l if l is not None else -math.inf for l in lower]
        hi_l = [u if u is not None else math.inf for u in upper]
        self._need_resident(value, "apply_thresholds")
        if n > 32:
            raise NotImplementedError("hm_apply_thresholds supports up to 32 channels on the last axis")
        value = value.contiguous()
        if self._std is not None and not self._std.is_contiguous():
            self._std = self._std.contiguous()
        self._eng().apply_thresholds_(value, self._std, lo_l, hi_l)          # hm_apply_thresholds, in place
        self.val = value

    def compute_dimension_statistics(self, axis=None):
        """modules/measurand.py:318-350."""
        values = self._f64()
        self._need_resident(values, "compute_dimension_statistics")
        if axis is None:                                                         # statistics over every element: one "channel"
            st = self._eng().channel_statistics(values.reshape(-1, 1), None if self._std is None else self._std.reshape(-1, 1))
            return {k_: (None if v is None else self._export(v.reshape(()))) for k_, v in st.items()}
        all_but_last = values.dim() >= 2 and \
            sorted(a % values.dim() for a in ((axis,) if isinstance(axis, int) else tuple(axis))) == list(range(values.dim() - 1))
        if all_but_last and values.shape[-1] <= 4:
            st = self._eng().channel_statistics(values, self._std)                  # hm_channel_statistics
        else:
            st = self._eng().axis_statistics(values, self._std, axis)               # hm_axis_statistics: any other axis / axis tuple
        return {k_: self._export(v) for k_, v in st.items()}

    def compute_kernel_density_estimate(self, data_points: int, included_range=None, channels=None, use_std: bool = False):
        """modules/measurand.py:716-761 is a NumPy-only plotting helper of the reference (scipy.stats.gaussian_kde on host arrays)
        and is out of scope here: this package computes on the device only."""
        raise NotImplementedError("kernel density estimates are host-side plotting support in the reference; "
                                  "use to_numpy() and scipy.stats.gaussian_kde")

    def compute_channel_histogram(self, bins: int, included_range=None, channels=None, use_std: bool = False):
        """modules/measurand.py:430-469: np.histogram per channel - hm_channel_histogram on the device (per-workgroup
        LDS histograms); up to 4 channels and bins * channels <= 8192."""
        if channels is None:
            channels = list(range(gs.NUM_OF_CHS))
        vd = self._f64()
        self._need_resident(vd, "compute_channel_histogram")
        if vd.shape[-1] > 4 or bins * vd.shape[-1] > 8192:
            raise NotImplementedError("hm_channel_histogram supports up to 4 channels and bins * channels <= 8192")
        return self._eng().channel_histogram(vd, self._std if use_std else None, bins, included_range, channels)

    @staticmethod
    def compute_difference(x: "HipMeasurand", y: "HipMeasurand", multiplier: float):
        """modules/measurand.py:620-655."""
        cls = x.__class__
        xv, yv = x._f64(), y._f64()
        x._need_resident(xv, "compute_difference")
        if not is_broadcastable(xv.shape, yv.shape):
            raise ValueError("Measurands are not broadcastable.")
        ad, ads, rd, rds = x._eng().compute_difference(xv, x._std, yv, y._std, multiplier)   # hm_compute_difference(_bcast)
        return cls(ad, ads), cls(rd, rds)

    @staticmethod
    def interpolate(x0: "HipMeasurand", x1: "HipMeasurand", y0: float, y1: float, y: float):
        """modules/measurand.py:657-681 (std formula as written)."""
        cls = x0.__class__
        x0._need_resident(x0._f64(), "interpolate")
        if not is_broadcastable(x0.shape, x1.shape):
            raise ValueError("Measurands are not broadcastable.")
        res, res_std = x0._eng().interpolate(x0._f64(), x0._std, x1._f64(), x1._std, y0, y1, y)   # hm_interpolate(_bcast)
        return cls(res, res_std)


def _keep(values: torch.Tensor, axis):
    axes = (axis,) if isinstance(axis, int) else tuple(axis)
    axes = tuple(a % values.dim() for a in axes)
    return tuple(1 if d in axes else values.shape[d] for d in range(values.dim()))


class _HostEngine:
    """engine.<function> with the host library active for the duration of the call (nat.host_mode())."""

    def __getattr__(self, name):
        from . import _native as nat
        fn = getattr(_engine(), name)
        if not callable(fn):
            return fn

        def call(*a, **k):
            with nat.host_mode():
                return fn(*a, **k)
        call.__name__ = name
        return call


_HOST_ENGINE = _HostEngine()


class HostMeasurand(HipMeasurand):
    """The reference's NumpyMeasurand slot (modules/measurand.py:684-714; `Measurand(use_cupy=False)`, modules/measurand_factory.py:10-14):
    `.val` / `.std` are NumPy arrays, `backend == "numpy"`, and every method computes in the HOST build of the same C ABI
    (csrc_host/hm_host.cpp -> lib/libhdrmerge_host.so: plain C++, OpenMP) - selected explicitly, never a fallback of the HIP backend.
    Internally the arrays are host torch tensors sharing memory with the NumPy arrays handed in and out (no copies)."""
    lib = np
    backend = "numpy"
    ArrayType = np.ndarray
    InputType = (int, float, np.ndarray, torch.Tensor)

    @classmethod
    def _device(cls) -> torch.device:
        return torch.device("cpu")

    @staticmethod
    def _export(t):
        return None if t is None else t.numpy()

    def _eng(self):
        return _HOST_ENGINE

    def _need_resident(self, t: torch.Tensor, what: str) -> None:
        if t.is_cuda:
            raise RuntimeError(f"{what}: a host Measurand holds host arrays")

    @classmethod
    def _to_tensor(cls, x, like: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        if x is None:
            return None
        if isinstance(x, (int, float)):
            return torch.tensor([x], dtype=_F64)
        if isinstance(x, np.ndarray):
            return torch.from_numpy(x if x.flags.c_contiguous and x.flags.writeable else np.ascontiguousarray(x))   # references stored, no copy (measurand.py:712-713)
        return x.cpu() if x.is_cuda else x
