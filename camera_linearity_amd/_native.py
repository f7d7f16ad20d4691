"""ctypes binding of libhdrmerge.so - the C ABI declared in include/hdrmerge.h.

There is no CPU fallback: if the library is missing, importing this module raises, and every
product entry point that needs it fails loudly. torch is imported first so that the HIP runtime the
library resolves (`libamdhip64.so.7`) is the one torch already loaded - device pointers and
streams are then shared between torch and the library.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib

import threading

import torch  # noqa: F401  (must precede the CDLL: see module docstring)

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("HDRMERGE_LIB", _HERE / "lib" / "libhdrmerge.so"))

HM_MAX_FRAMES = 32
HM_MAX_CHANNELS = 4
HM_MAX_DIMS = 6

HM_OK, HM_EINVAL, HM_EUNSUPPORTED, HM_EALIGN, HM_ELAUNCH, HM_ENODEVICE, HM_ESHAPE = 0, -1, -2, -3, -4, -5, -6
HM_OP_ADD, HM_OP_SUB, HM_OP_MUL, HM_OP_DIV, HM_OP_POW = range(5)
HM_UOP_NEG, HM_UOP_LOG_E, HM_UOP_LOG_10 = range(3)


class HdrMergeError(RuntimeError):
    """A libhdrmerge call returned an error that has no closer Python exception type."""


class MergeArgs(C.Structure):
    """struct hm_merge_args of include/hdrmerge.h (field order and types must match)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_frames", C.c_int32), ("channels", C.c_int32), ("variant", C.c_int32),
        ("height", C.c_int64), ("width", C.c_int64), ("row0", C.c_int64), ("rows", C.c_int64),
        ("buf_row0", C.c_int64), ("buf_rows", C.c_int64),
        ("frames_u8", C.POINTER(C.c_void_p)), ("frames_f64", C.POINTER(C.c_void_p)),
        ("stds", C.POINTER(C.c_void_p)), ("exposures", C.POINTER(C.c_double)),
        ("icrf", C.c_void_p), ("icrf_diff", C.c_void_p), ("w_lut", C.c_void_p), ("dw_lut", C.c_void_p),
        ("darks_u8", C.POINTER(C.c_void_p)), ("dark_min_dn", C.POINTER(C.c_int32)),
        ("median_k", C.c_int32), ("_pad0", C.c_int32),
        ("flat_u8", C.c_void_p), ("flat_f64", C.c_void_p), ("flat_std", C.c_void_p),
        ("ff_mean", C.c_double * HM_MAX_CHANNELS), ("ff_std_mean", C.c_double * HM_MAX_CHANNELS),
        ("out_val", C.c_void_p), ("out_std", C.c_void_p), ("out_sum_w", C.c_void_p),
        ("hot_workspace", C.c_void_p), ("hot_workspace_bytes", C.c_size_t),
        ("frames_workspace", C.c_void_p), ("frames_workspace_bytes", C.c_size_t),
    ]


_SIGNATURES = {
    "hm_version": (C.c_int, []),
    "hm_strerror": (C.c_char_p, [C.c_int]),
    "hm_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "hm_debug_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "hm_debug_copy_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p]),
    "hm_debug_stride_probe": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "hm_gaussian_weight_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_gaussian_weight_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_gaussian_weight_lut_host": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hm_u8_to_unit_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_linearize_u8": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "hm_linearize_f64": (C.c_int, [C.c_void_p] * 7 + [C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "hm_merge": (C.c_int, [C.POINTER(MergeArgs), C.c_void_p]),
    "hm_merge_hot_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "hm_merge_hot_workspace_min_bytes": (C.c_size_t, [C.c_int64]),
    "hm_merge_frames_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "hm_merge_algorithmic_bytes": (C.c_int64, [C.POINTER(MergeArgs)]),
    "hm_merge_describe": (C.c_int, [C.POINTER(MergeArgs), C.c_char_p, C.c_int]),
    "hm_hot_pixel_filter_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                         C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "hm_hot_pixel_filter_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                          C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "hm_roi_mean_workspace_bytes": (C.c_size_t, []),
    "hm_roi_mean_u8": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_roi_mean_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_normalize_by_map": (C.c_int, [C.c_void_p] * 5 + [C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                         C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "hm_binary_op": (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                              C.POINTER(C.c_int64), C.c_void_p]),
    "hm_unary_op": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_pow_scalar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_take_axis": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                               C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    "hm_apply_thresholds": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_int, C.c_void_p]),
    "hm_compute_difference": (C.c_int, [C.c_void_p] * 4 + [C.c_double] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p]),
    "hm_interpolate": (C.c_int, [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_pair_statistics_workspace_bytes": (C.c_size_t, []),
    "hm_pair_statistics": (C.c_int, [C.c_void_p] * 4 + [C.c_double, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_pairs_statistics_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hm_pairs_statistics": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_histogram_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hm_channel_minmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_channel_histogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_channel_statistics_workspace_bytes": (C.c_size_t, []),
    "hm_channel_statistics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_axis_statistics_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "hm_axis_statistics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hm_axis_statistics2_workspace_bytes": (C.c_size_t, [C.c_int64] * 5),
    "hm_axis_statistics2": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p] * 5),
    "hm_compute_difference_bcast": (C.c_int, [C.c_void_p] * 4 + [C.c_double] + [C.c_void_p] * 4 + [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                                                              C.POINTER(C.c_int64), C.c_void_p]),
    "hm_interpolate_bcast": (C.c_int, [C.c_void_p] * 4 + [C.c_double] * 3 + [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                                             C.POINTER(C.c_int64), C.c_void_p]),
    "hm_welford_update": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_int, C.c_void_p]),
    "hm_welford_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hm_welford_algorithmic_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int64]),
    "hm_tiff_lzw_decode": (C.c_int64, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]),
    "hm_tiff_packbits_decode": (C.c_int64, [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]),
    "hm_linearity_energy_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "hm_linearity_energy": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class _Lib:
    """The loaded library with a per-symbol call counter (`lib.calls["hm_merge"]`): tests use it to assert that an
    operation ran through the HIP kernel it is documented to use, not through some other path."""

    def __init__(self, cdll):
        import collections
        self._cdll = cdll
        self.calls = collections.Counter()

    def _bind(self, name, res, args):
        fn = getattr(self._cdll, name)          # AttributeError if the library lacks a declared entry point
        fn.restype = res
        fn.argtypes = args
        calls = self.calls

        def call(*a):
            calls[name] += 1
            return fn(*a)
        call.__name__ = name
        setattr(self, name, call)


def _load():
    if not LIB_PATH.exists():
        raise ImportError(
            f"libhdrmerge.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C camera_linearity_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    lib = _Lib(C.CDLL(str(LIB_PATH)))
    for name, (res, args) in _SIGNATURES.items():
        lib._bind(name, res, args)
    return lib


HM_ABI_VERSION = 2
hip_lib = _load()
if hip_lib.hm_version() != HM_ABI_VERSION:
    raise ImportError(f"libhdrmerge ABI version {hip_lib.hm_version()} != {HM_ABI_VERSION} expected by this package")

# ---- the HOST build of the same ABI (csrc_host/hm_host.cpp -> lib/libhdrmerge_host.so): the reference's NumPy slot, Measurand(use_cupy=False).
# Loaded on first use and only through host_mode(); the HIP path never reaches it (engine._require_cuda rejects host tensors outside host_mode).
HOST_LIB_PATH = pathlib.Path(os.environ.get("HDRMERGE_HOST_LIB", _HERE / "lib" / "libhdrmerge_host.so"))
HOST_ONLY_MISSING = ("hm_debug_clock_probe", "hm_debug_copy_probe", "hm_debug_stride_probe", "hm_tiff_lzw_decode", "hm_tiff_packbits_decode")
_host_lib = None
_mode = threading.local()


def host_lib():
    global _host_lib
    if _host_lib is None:
        if not HOST_LIB_PATH.exists():
            raise ImportError(f"libhdrmerge_host.so not found at {HOST_LIB_PATH}: build it with `make -C camera_linearity_amd/csrc host` (g++)")
        h = _Lib(C.CDLL(str(HOST_LIB_PATH)))
        for name, (res, args) in _SIGNATURES.items():
            if name not in HOST_ONLY_MISSING:
                h._bind(name, res, args)
        if h.hm_version() != HM_ABI_VERSION:
            raise ImportError(f"libhdrmerge_host ABI version {h.hm_version()} != {HM_ABI_VERSION}")
        if "OMP_NUM_THREADS" not in os.environ:            # OpenMP would start one thread per core of the machine: keep to a one-GPU box's CPU share
            try:
                C.CDLL("libgomp.so.1").omp_set_num_threads(min(16, os.cpu_count() or 1))
            except OSError:
                pass
        _host_lib = h
    return _host_lib


def in_host_mode() -> bool:
    return getattr(_mode, "depth", 0) > 0


class host_mode:
    """Context manager: inside it `lib` is the host build and engine functions take HOST tensors (and only those). Entered by
    HostMeasurand / host ImageSets around every engine call; never entered by the HIP backend."""

    def __enter__(self):
        host_lib()
        _mode.depth = getattr(_mode, "depth", 0) + 1
        return self

    def __exit__(self, *exc):
        _mode.depth -= 1
        return False


class _Dispatch:
    """`lib`: the HIP library, or - inside host_mode() - the host library. `calls` is the active library's call counter."""

    def __getattr__(self, name):
        return getattr(_host_lib if in_host_mode() else hip_lib, name)


lib = _Dispatch()


def strerror(code: int) -> str:
    return lib.hm_strerror(code).decode()


def check(code: int, what: str = "libhdrmerge") -> None:
    """Map hm_* status codes to the exception types the reference raises for the same misuse
    (ValueError for shape/argument problems, modules/measurand.py:112,404,710)."""
    if code == HM_OK:
        return
    msg = f"{what}: {strerror(code)} ({code})"
    if code in (HM_EINVAL, HM_ESHAPE, HM_EALIGN):
        raise ValueError(msg)
    if code == HM_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise HdrMergeError(msg)


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None stays NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
