"""Functional layer over the C ABI: torch device tensors in, torch device tensors out.

Everything here is plumbing - argument checking, pointer tables, output allocation - around the
hand-written HIP kernels in csrc/. No arithmetic of the hot path happens in Python or in torch ops.
The only host-side numbers are the 256-entry Gaussian weight tables (modules/measurand.py:615-616
evaluated on the DN grid with NumPy, so that they are bit-identical to what the reference computes
for 8-bit frames) and per-frame scalars (1/exposure, DN thresholds).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native as nat
from .settings import BITS, MAX_DN

_F64 = torch.float64
_U8 = torch.uint8


# ---------------------------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------------------------
def _require_cuda(t: torch.Tensor, name: str) -> None:
    """The operand lives where the active library computes: in HBM for the HIP backend - or, inside nat.host_mode() (the explicit
    host backend, Measurand(use_cupy=False)), in host memory. The HIP backend never takes host tensors: there is no CPU fallback."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)} instead.")
    if nat.in_host_mode():
        if t.is_cuda:
            raise RuntimeError(f"{name} lives on {t.device}; the host backend computes on host tensors")
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name} lives on {t.device}; the HIP backend only computes on device tensors "
                           "(there is no CPU fallback)")


class _NoDevice:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _on(device):
    """torch.cuda.device(device) for the HIP backend, nothing for host tensors."""
    return _NoDevice() if torch.device(device).type != "cuda" else torch.cuda.device(device)


def _dev_f64(x, device) -> torch.Tensor:
    """float64 contiguous tensor on `device` from a tensor / ndarray / sequence (tables, not images)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=_F64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float64)), device=device)


def _ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


_weight_cache: Dict[str, tuple] = {}


def weight_luts_host():
    """w, dw on the DN grid v = k/255: modules/measurand.py:615-616 with NumPy (`np.e ** x`)."""
    v = np.arange(BITS, dtype=np.uint8).astype(np.float64) / MAX_DN       # modules/image_set.py:223
    w = np.e ** (-30 * (v - 0.5) ** 2)
    dw = -2 * 30 * (v - 0.5) * w
    return w, dw


def weight_luts(device) -> tuple:
    key = str(torch.device(device))
    if key not in _weight_cache:
        w, dw = weight_luts_host()
        _weight_cache[key] = (torch.as_tensor(w, device=device), torch.as_tensor(dw, device=device))
    return _weight_cache[key]


def dark_min_dn(scale: float, threshold: float) -> int:
    """Smallest dark DN that counts as hot: the reference tests `dark.val > threshold`
    (modules/measurand.py:545) with dark.val = (DN/255) * scale (modules/image_set.py:223,260);
    evaluated here for all 256 DNs with the same float64 operations, so the DN-domain comparison the
    kernel performs (`dark_dn >= min_dn`) is exactly equivalent. Returns 256 when no DN is hot."""
    v = np.arange(BITS, dtype=np.uint8).astype(np.float64) / MAX_DN
    if scale != 1.0:
        v = np.array([scale], dtype=np.float64) * v                       # Measurand(scale) * dark, measurand.py:198
    hot = np.nonzero(v > threshold)[0]
    return int(hot[0]) if hot.size else BITS


def _stream(device):
    return None if torch.device(device).type != "cuda" else nat.current_stream_ptr(device)


# ---------------------------------------------------------------------------------------------
# rows 1, 3, 4: per-frame kernels
# ---------------------------------------------------------------------------------------------
def u8_to_unit(dn: torch.Tensor) -> torch.Tensor:
    """modules/image_set.py:223."""
    _require_cuda(dn, "dn")
    dn = dn.contiguous()
    out = torch.empty(dn.shape, dtype=_F64, device=dn.device)
    with _on(dn.device):
        nat.check(nat.lib.hm_u8_to_unit_f64(dn.data_ptr(), out.data_ptr(), dn.numel(), _stream(dn.device)), "hm_u8_to_unit_f64")
    return out


def gaussian_weight(val: torch.Tensor):
    """modules/measurand.py:606-618 -> (w, dw)."""
    _require_cuda(val, "val")
    val = val.contiguous()
    w = torch.empty(val.shape, dtype=_F64, device=val.device)
    dw = torch.empty(val.shape, dtype=_F64, device=val.device)
    with _on(val.device):
        if val.dtype == _U8:
            wl, dwl = weight_luts(val.device)
            rc = nat.lib.hm_gaussian_weight_u8(val.data_ptr(), wl.data_ptr(), dwl.data_ptr(), w.data_ptr(), dw.data_ptr(),
                                               val.numel(), _stream(val.device))
        elif val.dtype == _F64:
            rc = nat.lib.hm_gaussian_weight_f64(val.data_ptr(), w.data_ptr(), dw.data_ptr(), val.numel(), _stream(val.device))
        else:
            raise TypeError(f"gaussian_weight expects uint8 or float64, got {val.dtype}")
    nat.check(rc, "hm_gaussian_weight")
    return w, dw


def linearize(val: torch.Tensor, std: Optional[torch.Tensor], icrf, icrf_diff=None, return_index: bool = False):
    """modules/measurand.py:471-541. `icrf` is (256, C) or 1-D (256,). Returns (val, std|None[, idx])."""
    _require_cuda(val, "val")
    dev = val.device
    val = val.contiguous()
    icrf_t = _dev_f64(icrf, dev)
    diff_t = None if icrf_diff is None else _dev_f64(icrf_diff, dev)
    C_ = val.shape[-1] if val.dim() > 0 else 1
    if icrf_t.dim() == 1:
        lut_stride = 1
    else:
        if icrf_t.shape[-1] != C_:
            raise ValueError(f"ICRF has {icrf_t.shape[-1]} channels, value has {C_}")
        lut_stride = C_
    if icrf_t.shape[0] != BITS:
        raise ValueError(f"ICRF must have {BITS} rows, got {icrf_t.shape[0]}")
    use_std = std is not None and diff_t is not None                      # measurand.py:498-500
    if use_std:
        _require_cuda(std, "std")
        if std.shape != val.shape:
            raise ValueError("Value and std shapes must match.")
        std = std.to(_F64).contiguous()
    out_val = torch.empty(val.shape, dtype=_F64, device=dev)
    out_std = torch.empty(val.shape, dtype=_F64, device=dev) if use_std else None
    idx = None
    with _on(dev):
        if val.dtype == _U8:
            rc = nat.lib.hm_linearize_u8(val.data_ptr(), nat.ptr(std) if use_std else None, icrf_t.data_ptr(), nat.ptr(diff_t),
                                         out_val.data_ptr(), nat.ptr(out_std), val.numel(), C_, lut_stride, _stream(dev))
            if return_index:
                idx = val.clone()                                          # measurand.py:505
        elif val.dtype == _F64:
            if return_index:
                idx = torch.empty(val.shape, dtype=_U8, device=dev)
            rc = nat.lib.hm_linearize_f64(val.data_ptr(), nat.ptr(std) if use_std else None, icrf_t.data_ptr(), nat.ptr(diff_t),
                                          out_val.data_ptr(), nat.ptr(out_std), nat.ptr(idx), val.numel(), C_, lut_stride,
                                          _stream(dev))
        else:
            raise TypeError(f"linearize expects uint8 or float64 values, got {val.dtype}")
    nat.check(rc, "hm_linearize")
    return (out_val, out_std, idx) if return_index else (out_val, out_std)


# ---------------------------------------------------------------------------------------------
# rows 8, 9 standalone
# ---------------------------------------------------------------------------------------------
def hot_pixel_filter(x: torch.Tensor, dark_map: torch.Tensor, threshold: float, median_k: int,
                     min_dn: Optional[int] = None) -> torch.Tensor:
    """modules/measurand.py:543-557 (intended semantics). `dark_map` is a uint8 DN map (hot iff
    DN >= min_dn; min_dn defaults to dark_min_dn(1.0, threshold)) or a float64 value map (hot iff > threshold)."""
    _require_cuda(x, "x")
    _require_cuda(dark_map, "dark_map")
    if x.dim() != 3 or dark_map.shape != x.shape:
        raise ValueError("hot_pixel_filter expects (H, W, C) arrays of equal shape")
    x = x.contiguous()
    dark_map = dark_map.contiguous()
    H, W, Cc = x.shape
    out = torch.empty_like(x)
    mu8 = dark_map.data_ptr() if dark_map.dtype == _U8 else None
    mf64 = dark_map.data_ptr() if dark_map.dtype == _F64 else None
    if mu8 is None and mf64 is None:
        raise TypeError("dark map must be uint8 or float64")
    if min_dn is None:
        min_dn = dark_min_dn(1.0, threshold)
    fn = {_U8: nat.lib.hm_hot_pixel_filter_u8, _F64: nat.lib.hm_hot_pixel_filter_f64}.get(x.dtype)
    if fn is None:
        raise TypeError("hot_pixel_filter expects uint8 or float64 data")
    with _on(x.device):
        nat.check(fn(x.data_ptr(), mu8, mf64, int(min_dn), float(threshold), int(median_k), out.data_ptr(), H, W, Cc,
                     _stream(x.device)), "hm_hot_pixel_filter")
    return out


def flat_roi_bounds(size_x: int, size_y: int, p: float):
    """modules/measurand.py:570-576 with the integer ROI index."""
    import math
    dx = math.floor(size_x * p)
    dy = math.floor(size_y * p)
    i = (math.floor(1 / p) - 1) // 2
    return i * dx, (i + 1) * dx, i * dy, (i + 1) * dy


def roi_mean(img: torch.Tensor, x0: int, x1: int, y0: int, y1: int) -> torch.Tensor:
    """modules/measurand.py:579 - per-channel mean over img[x0:x1, y0:y1, :] (uint8 images are DN/255)."""
    _require_cuda(img, "img")
    if img.dim() != 3:
        raise ValueError("roi_mean expects an (H, W, C) image")
    img = img.contiguous()
    H, W, Cc = img.shape
    out = torch.empty(Cc, dtype=_F64, device=img.device)
    ws = torch.empty(nat.lib.hm_roi_mean_workspace_bytes() // 8, dtype=_F64, device=img.device)
    fn = {_U8: nat.lib.hm_roi_mean_u8, _F64: nat.lib.hm_roi_mean_f64}.get(img.dtype)
    if fn is None:
        raise TypeError("roi_mean expects uint8 or float64")
    with _on(img.device):
        nat.check(fn(img.data_ptr(), H, W, Cc, x0, x1, y0, y1, out.data_ptr(), ws.data_ptr(), _stream(img.device)), "hm_roi_mean")
    return out


def normalize_by_map(val: torch.Tensor, std: Optional[torch.Tensor], flat: torch.Tensor, flat_std: Optional[torch.Tensor],
                     ff_mean, ff_std_mean=None):
    """modules/measurand.py:585-604. `flat` uint8 DN or float64; means are host sequences of C floats."""
    _require_cuda(val, "val")
    val = val.contiguous()
    flat = flat.contiguous()
    Cc = val.shape[-1]
    m = (C.c_double * nat.HM_MAX_CHANNELS)(*[float(x) for x in ff_mean])
    s = (C.c_double * nat.HM_MAX_CHANNELS)(*[float(x) for x in (ff_std_mean if ff_std_mean is not None else [0.0] * Cc)])
    out_val = torch.empty_like(val)
    out_std = torch.empty_like(val) if std is not None else None
    with _on(val.device):
        nat.check(nat.lib.hm_normalize_by_map(
            val.data_ptr(), nat.ptr(std.contiguous()) if std is not None else None,
            flat.data_ptr() if flat.dtype == _U8 else None, flat.data_ptr() if flat.dtype == _F64 else None,
            nat.ptr(flat_std.contiguous()) if flat_std is not None else None,
            C.cast(m, C.POINTER(C.c_double)), C.cast(s, C.POINTER(C.c_double)),
            out_val.data_ptr(), nat.ptr(out_std), val.numel(), Cc, _stream(val.device)), "hm_normalize_by_map")
    return out_val, out_std


# ---------------------------------------------------------------------------------------------
# rows 5-9 fused: the merge
# ---------------------------------------------------------------------------------------------
class MergePlan:
    """A validated hm_merge_args plus the tensors it points into (kept alive until the plan dies).
    `launch()` enqueues the fused kernel on the current stream of the plan's device; it can be called
    repeatedly (bench, hipGraph capture)."""

    def __init__(self, args: nat.MergeArgs, keep: list, device, outputs: dict):
        self.args = args
        self._keep = keep
        self.device = torch.device(device)
        self.outputs = outputs
        self._ref = C.byref(args)
        self.host = self.device.type != "cuda"             # a plan of the host backend (built inside nat.host_mode())

    def launch(self, stream: Optional[int] = None) -> None:
        if self.host:
            with nat.host_mode():
                nat.check(nat.lib.hm_merge(self._ref, None), "hm_merge")
            return
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if torch.cuda.current_device() == self.device.index:
            rc = nat.lib.hm_merge(self._ref, stream)
        else:
            with _on(self.device):
                rc = nat.lib.hm_merge(self._ref, stream)
        if rc:
            nat.check(rc, "hm_merge")

    @property
    def algorithmic_bytes(self) -> int:
        return int(nat.hip_lib.hm_merge_algorithmic_bytes(C.byref(self.args)))

    @property
    def kernels(self) -> str:
        """Names of the kernels launch() dispatches to, in launch order (hm_merge_describe: the library's own dispatch, dry)."""
        buf = C.create_string_buffer(512)
        lib = nat.host_lib() if self.host else nat.hip_lib
        nat.check(lib.hm_merge_describe(self._ref, buf, 512), "hm_merge_describe")
        return buf.value.decode()


def plan_merge(frames: Sequence[torch.Tensor], exposures: Sequence[float], icrf, icrf_diff=None,
               stds: Optional[Sequence[torch.Tensor]] = None,
               darks: Optional[Sequence[Optional[torch.Tensor]]] = None,
               dark_min: Optional[Sequence[int]] = None, median_k: int = 3,
               flat: Optional[torch.Tensor] = None, flat_std: Optional[torch.Tensor] = None,
               ff_mean=None, ff_std_mean=None, want_sum_w: bool = False, want_val: bool = True,
               height: Optional[int] = None, row0: int = 0, rows: Optional[int] = None, buf_row0: int = 0,
               variant: int = 0, hot_queue: bool = True) -> MergePlan:
    """Build the launch descriptor for one fused merge (modules/exposure_series.py:317-419).

    frames : N tensors (buf_rows, W, C), all uint8 DNs or all float64 values, ascending exposure.
             They cover image rows [buf_row0, buf_row0 + buf_rows) of an image `height` rows tall;
             the call produces rows [row0, row0 + rows). Defaults: the buffers are the whole image.
    darks  : per frame a uint8 dark DN map covering the same rows as the frames, or None;
             dark_min[i] = smallest hot DN (see dark_min_dn()). hot_queue: give the hot-pixel pass its queue workspace
             (hm_merge_hot_workspace_bytes: 1 byte per output element) so that it stays balanced on dense maps;
             False = the workspace-free path (one hot element per wave at a time; sparse maps only).
    flat, flat_std : flat-field value (uint8 DN or float64) and float64 uncertainty covering the
             OUTPUT rows; ff_mean / ff_std_mean are the C ROI means (host floats).
    """
    n = len(frames)
    if n == 0:
        raise ValueError("merge needs at least one frame")
    for i, f in enumerate(frames):
        _require_cuda(f, f"frames[{i}]")
    dev = frames[0].device
    dt = frames[0].dtype
    if dt not in (_U8, _F64):
        raise TypeError(f"frames must be uint8 or float64, got {dt}")
    shape = tuple(frames[0].shape)
    if len(shape) != 3:
        raise ValueError(f"frames must be (H, W, C), got shape {shape}")
    keep: list = []
    fr = []
    for i, f in enumerate(frames):
        if f.device != dev or f.dtype != dt or tuple(f.shape) != shape:
            raise ValueError("all frames must share device, dtype and shape")
        f = f.contiguous()
        fr.append(f)
    keep.extend(fr)
    buf_rows, W, Cc = shape
    height = buf_rows + buf_row0 if height is None else int(height)
    rows = (buf_row0 + buf_rows - row0) if rows is None else int(rows)
    if len(exposures) != n:
        raise ValueError("one exposure per frame is required")
    with_std = stds is not None
    sd = []
    if with_std:
        if len(stds) != n:
            raise ValueError("one std frame per value frame is required")
        if icrf_diff is None:
            raise ValueError("ICRF_diff is required to propagate uncertainty")
        for i, s in enumerate(stds):
            _require_cuda(s, f"stds[{i}]")
            if tuple(s.shape) != shape:
                raise ValueError("Value and std shapes must match.")
            sd.append(s.to(_F64).contiguous())
        keep.extend(sd)
    icrf_t = _dev_f64(icrf, dev)
    if tuple(icrf_t.shape) != (BITS, Cc):
        raise ValueError(f"ICRF must have shape ({BITS}, {Cc}), got {tuple(icrf_t.shape)}")
    diff_t = None
    if icrf_diff is not None:
        diff_t = _dev_f64(icrf_diff, dev)
        if tuple(diff_t.shape) != (BITS, Cc):
            raise ValueError(f"ICRF_diff must have shape ({BITS}, {Cc})")
    w_lut, dw_lut = weight_luts(dev)
    keep += [icrf_t, diff_t, w_lut, dw_lut]

    a = nat.MergeArgs()
    a.struct_size = C.sizeof(nat.MergeArgs)
    a.n_frames, a.channels, a.variant = n, Cc, int(variant)
    a.height, a.width, a.row0, a.rows, a.buf_row0, a.buf_rows = height, W, int(row0), rows, int(buf_row0), buf_rows
    fptrs = _ptr_array(fr)
    if dt == _U8:
        a.frames_u8 = C.cast(fptrs, C.POINTER(C.c_void_p))
    else:
        a.frames_f64 = C.cast(fptrs, C.POINTER(C.c_void_p))
    keep.append(fptrs)
    if with_std:
        sptrs = _ptr_array(sd)
        a.stds = C.cast(sptrs, C.POINTER(C.c_void_p))
        keep.append(sptrs)
    exp = (C.c_double * n)(*[float(t) for t in exposures])
    a.exposures = C.cast(exp, C.POINTER(C.c_double))
    keep.append(exp)
    a.icrf, a.icrf_diff = icrf_t.data_ptr(), nat.ptr(diff_t)
    a.w_lut, a.dw_lut = w_lut.data_ptr(), dw_lut.data_ptr()
    if darks is not None and any(d is not None for d in darks):
        if len(darks) != n or dark_min is None or len(dark_min) != n:
            raise ValueError("darks and dark_min need one entry per frame")
        dk = []
        for i, d in enumerate(darks):
            if d is None:
                dk.append(None)
                continue
            _require_cuda(d, f"darks[{i}]")
            if d.dtype != _U8 or tuple(d.shape) != shape:
                raise ValueError("dark maps must be uint8 with the frames' shape")
            dk.append(d.contiguous())
        dptrs = _ptr_array(dk)
        dmin = (C.c_int32 * n)(*[int(x) for x in dark_min])
        a.darks_u8 = C.cast(dptrs, C.POINTER(C.c_void_p))
        a.dark_min_dn = C.cast(dmin, C.POINTER(C.c_int32))
        a.median_k = int(median_k)
        keep += [dk, dptrs, dmin]
        if hot_queue:
            ws_bytes = int(nat.lib.hm_merge_hot_workspace_bytes(rows * W * Cc))
            ws = torch.empty(ws_bytes, dtype=_U8, device=dev)          # torch allocations are 512-byte aligned
            a.hot_workspace, a.hot_workspace_bytes = ws.data_ptr(), ws_bytes
            keep.append(ws)
    out_shape = (rows, W, Cc)
    if flat is not None:
        _require_cuda(flat, "flat")
        if tuple(flat.shape) != out_shape:
            raise ValueError(f"flat field must cover the output rows: expected {out_shape}, got {tuple(flat.shape)}")
        flat = flat.contiguous()
        if flat.dtype == _U8:
            a.flat_u8 = flat.data_ptr()
        elif flat.dtype == _F64:
            a.flat_f64 = flat.data_ptr()
        else:
            raise TypeError("flat must be uint8 or float64")
        if ff_mean is None:
            raise ValueError("ff_mean is required with a flat field")
        for c in range(Cc):
            a.ff_mean[c] = float(ff_mean[c])
        keep.append(flat)
        if with_std:
            if flat_std is None or ff_std_mean is None:
                raise ValueError("flat_std and ff_std_mean are required to propagate uncertainty through the flat field")
            _require_cuda(flat_std, "flat_std")
            flat_std = flat_std.to(_F64).contiguous()
            if tuple(flat_std.shape) != out_shape:
                raise ValueError("flat_std must cover the output rows")
            a.flat_std = flat_std.data_ptr()
            for c in range(Cc):
                a.ff_std_mean[c] = float(ff_std_mean[c])
            keep.append(flat_std)
    outputs = {}
    if want_val:
        outputs["val"] = torch.empty(out_shape, dtype=_F64, device=dev)
        a.out_val = outputs["val"].data_ptr()
        if with_std:
            outputs["std"] = torch.empty(out_shape, dtype=_F64, device=dev)
            a.out_std = outputs["std"].data_ptr()
    if want_sum_w:
        outputs["sum_w"] = torch.empty(out_shape, dtype=_F64, device=dev)
        a.out_sum_w = outputs["sum_w"].data_ptr()
    # more than HM_MAX_FRAMES frames (or a forced chunking, variant <= -2): hm_merge runs HM_MAX_FRAMES frames per launch and keeps the
    # running sum of weights in out_sum_w, or in this workspace when the call has none
    fw = int(nat.lib.hm_merge_frames_workspace_bytes(n if variant > -2 else nat.HM_MAX_FRAMES + 1, rows * W * Cc, int(want_sum_w)))
    if fw:
        ws = torch.empty(fw, dtype=_U8, device=dev)
        a.frames_workspace, a.frames_workspace_bytes = ws.data_ptr(), fw
        keep.append(ws)
    return MergePlan(a, keep, dev, outputs)


def merge(frames, exposures, icrf, icrf_diff=None, stds=None, **kw) -> dict:
    """plan_merge(...).launch(); returns {'val', 'std'?, 'sum_w'?} device tensors."""
    plan = plan_merge(frames, exposures, icrf, icrf_diff, stds, **kw)
    plan.launch()
    return plan.outputs


class PlanGraph:
    """A sequence of merge plans recorded ONCE into a hipGraph and replayed with one host call.

    The library makes no synchronising call, allocates nothing and copies nothing (every entry point only enqueues kernels on the stream it
    is given; workspaces are the caller's), so whatever `MergePlan.launch()` enqueues - the streaming kernel, the dark-map scan and patch,
    the launches of a chunked stack - is capturable as it is. What a graph saves is host time: ~6 us of ctypes + argument checking +
    kernel-argument packing per hm_merge call, which is the whole cost of a small stack (BASELINE config 1, 3 x 256 x 256 x 3: the kernel
    runs for ~3 us). Streams of LARGE stacks gain nothing (the device is busy for longer than the host needs to enqueue the next launch).

    The graph holds the plans (and through them every tensor the recorded kernels read or write): inputs are refreshed by copying into
    the plans' frame tensors (`frames[i].copy_(...)`), outputs are read from `plans[i].outputs` after `replay()`.
    """

    def __init__(self, plans: Sequence[MergePlan], warmup: bool = True, lanes: int = 1):
        """lanes > 1: the plans are dealt round-robin onto `lanes` parallel branches of the graph (forked from and joined to the capture
        stream), so that kernels too small to fill 256 CUs overlap; the plans must then be independent of each other (no plan reads what
        another writes)."""
        plans = list(plans)
        if not plans:
            raise ValueError("PlanGraph needs at least one plan")
        if any(p.host for p in plans):
            raise TypeError("PlanGraph records device launches; host-backend plans run synchronously in launch()")
        dev = plans[0].device
        if any(p.device != dev for p in plans):
            raise ValueError("all plans of a graph must live on one device")
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.plans = plans
        self.device = dev
        self.lanes = min(int(lanes), len(plans))
        with torch.cuda.device(dev):
            side = torch.cuda.Stream(dev)
            branches = [torch.cuda.Stream(dev) for _ in range(self.lanes - 1)]
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                if warmup:                      # first launches query the kernels' occupancy and load their code objects: outside the capture
                    for p in plans:
                        p.launch()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=side):
                    for b in branches:          # fork
                        b.wait_stream(side)
                    for i, p in enumerate(plans):
                        lane = i % self.lanes
                        p.launch(side.cuda_stream if lane == 0 else branches[lane - 1].cuda_stream)
                    for b in branches:          # join
                        side.wait_stream(b)
            torch.cuda.current_stream(dev).wait_stream(side)

    def replay(self) -> None:
        """Enqueue the recorded launches on the current stream of the graph's device."""
        if torch.cuda.current_device() == self.device.index:
            self.graph.replay()
        else:
            with torch.cuda.device(self.device):
                self.graph.replay()


def sum_of_weights(frames, darks=None, dark_min=None, median_k: int = 3, **kw):
    """modules/exposure_series.py:317-345 -> (S, S**2) as device tensors."""
    Cc = frames[0].shape[-1]
    ident = np.zeros((BITS, Cc))
    plan = plan_merge(frames, [1.0] * len(frames), ident, darks=darks, dark_min=dark_min, median_k=median_k,
                      want_sum_w=True, want_val=False, **kw)
    plan.launch()
    S = plan.outputs["sum_w"]
    return S, elementwise_binary(nat.HM_OP_MUL, S, None, S, None)[0]


# ---------------------------------------------------------------------------------------------
# row 10: operators
# ---------------------------------------------------------------------------------------------
def _bcast_strides(t: torch.Tensor, shape) -> List[int]:
    pad = len(shape) - t.dim()
    st = [0] * pad + list(t.stride())
    sh = [1] * pad + list(t.shape)
    return [0 if sh[d] == 1 and shape[d] != 1 else st[d] for d in range(len(shape))]


def elementwise_binary(op: int, x1: torch.Tensor, s1, x2: torch.Tensor, s2):
    """modules/measurand.py:106-241 for two float64 device operands (NumPy broadcasting)."""
    _require_cuda(x1, "x1")
    _require_cuda(x2, "x2")
    x1 = x1.to(_F64).contiguous()
    x2 = x2.to(_F64).contiguous()
    if s1 is not None:
        s1 = s1.to(_F64).contiguous()
    if s2 is not None:
        s2 = s2.to(_F64).contiguous()
    try:
        shape = torch.broadcast_shapes(x1.shape, x2.shape)
    except RuntimeError:
        raise ValueError("Measurands are not broadcastable.")
    if len(shape) > nat.HM_MAX_DIMS:
        raise NotImplementedError(f"more than {nat.HM_MAX_DIMS} dimensions")
    if len(shape) == 0:
        shape = (1,)
    nd = len(shape)
    out = torch.empty(shape, dtype=_F64, device=x1.device)
    with_std = s1 is not None or s2 is not None
    out_std = torch.empty(shape, dtype=_F64, device=x1.device) if with_std else None
    sh = (C.c_int64 * nd)(*shape)
    st1 = (C.c_int64 * nd)(*_bcast_strides(x1, shape))
    st2 = (C.c_int64 * nd)(*_bcast_strides(x2, shape))
    with _on(x1.device):
        nat.check(nat.lib.hm_binary_op(op, x1.data_ptr(), nat.ptr(s1), x2.data_ptr(), nat.ptr(s2), out.data_ptr(),
                                       nat.ptr(out_std), nd, sh, st1, st2, _stream(x1.device)), "hm_binary_op")
    return out, out_std


def pow_scalar(x: torch.Tensor, s: Optional[torch.Tensor], exponent: float):
    """modules/measurand.py:217-241 for a plain scalar exponent -> (val, std | None); hm_pow_scalar (no pow() for 2, 0.5, 1, small integers)."""
    _require_cuda(x, "val")
    x = x.contiguous()
    s = None if s is None else s.to(_F64).contiguous()
    out = torch.empty_like(x)
    out_s = None if s is None else torch.empty_like(x)
    with _on(x.device):
        nat.check(nat.lib.hm_pow_scalar(x.data_ptr(), nat.ptr(s), float(exponent), out.data_ptr(), nat.ptr(out_s), x.numel(),
                                        _stream(x.device)), "hm_pow_scalar")
    return out, out_s


def elementwise_unary(op: int, x: torch.Tensor, s):
    _require_cuda(x, "x")
    x = x.to(_F64).contiguous()
    if s is not None:
        s = s.to(_F64).contiguous()
    out = torch.empty_like(x)
    out_std = torch.empty_like(x) if s is not None else None
    with _on(x.device):
        nat.check(nat.lib.hm_unary_op(op, x.data_ptr(), nat.ptr(s), out.data_ptr(), nat.ptr(out_std), x.numel(),
                                      _stream(x.device)), "hm_unary_op")
    return out, out_std


# ---------------------------------------------------------------------------------------------
# linearity statistics
# ---------------------------------------------------------------------------------------------
HM_TAKE_MAX = 16          # include/hdrmerge.h


def take_axis(x: torch.Tensor, s: Optional[torch.Tensor], indices: Sequence[int], axis: Optional[int]):
    """modules/measurand.py:352-373 (`lib.take(val, dims, axis)`): axis=None indexes the flattened array."""
    _require_cuda(x, "val")
    x = x.contiguous()
    if s is not None:
        _require_cuda(s, "std")
        s = s.to(_F64).contiguous()
    idx = [int(i) for i in indices]
    if axis is None:
        outer, axis_len, inner, out_shape = 1, x.numel(), 1, (len(idx),)
    else:
        ax = axis % x.dim()
        outer = int(np.prod(x.shape[:ax], dtype=np.int64))
        inner = int(np.prod(x.shape[ax + 1:], dtype=np.int64))
        axis_len = x.shape[ax]
        out_shape = tuple(x.shape[:ax]) + (len(idx),) + tuple(x.shape[ax + 1:])
    if any(i < -axis_len or i >= axis_len for i in idx):
        raise IndexError(f"index out of bounds for axis of size {axis_len}")
    out = torch.empty(out_shape, dtype=_F64, device=x.device)
    out_s = None if s is None else torch.empty(out_shape, dtype=_F64, device=x.device)
    with _on(x.device):
        if len(idx) <= HM_TAKE_MAX:
            arr = (C.c_int64 * len(idx))(*idx)
            nat.check(nat.lib.hm_take_axis(x.data_ptr(), nat.ptr(s), out.data_ptr(), nat.ptr(out_s), outer, axis_len, inner,
                                           arr, len(idx), _stream(x.device)), "hm_take_axis")
        else:
            # np.take has no limit on the number of indices; hm_take_axis takes HM_TAKE_MAX per launch: one launch per chunk into its
            # slice of the dense (outer, n, inner) output (a strided device copy when outer > 1)
            ov = out.view(outer, len(idx), inner)
            osv = None if out_s is None else out_s.view(outer, len(idx), inner)
            for k0 in range(0, len(idx), HM_TAKE_MAX):
                part = idx[k0:k0 + HM_TAKE_MAX]
                arr = (C.c_int64 * len(part))(*part)
                direct = outer == 1
                tv = ov[:, k0:k0 + len(part)] if direct else torch.empty((outer, len(part), inner), dtype=_F64, device=x.device)
                ts = None if s is None else (osv[:, k0:k0 + len(part)] if direct else torch.empty_like(tv))
                nat.check(nat.lib.hm_take_axis(x.data_ptr(), nat.ptr(s), tv.data_ptr(), nat.ptr(ts), outer, axis_len, inner,
                                               arr, len(part), _stream(x.device)), "hm_take_axis")
                if not direct:
                    ov[:, k0:k0 + len(part)].copy_(tv)
                    if ts is not None:
                        osv[:, k0:k0 + len(part)].copy_(ts)
    return out, out_s


def apply_thresholds_(val: torch.Tensor, std: Optional[torch.Tensor], lower: Sequence[float], upper: Sequence[float]) -> None:
    """modules/measurand.py:375-428, in place on contiguous float64 device tensors (last axis = channels)."""
    _require_cuda(val, "val")
    Cc = val.shape[-1]
    lo = (C.c_double * Cc)(*[float(x) for x in lower])
    hi = (C.c_double * Cc)(*[float(x) for x in upper])
    with _on(val.device):
        nat.check(nat.lib.hm_apply_thresholds(val.data_ptr(), nat.ptr(std), lo, hi, val.numel(), Cc, _stream(val.device)),
                  "hm_apply_thresholds")


def _bcast_setup(a, sa, b, sb):
    """Operands of compute_difference / interpolate that do not have one shape: the broadcast result shape and the element strides
    (0 on broadcast axes) of the two contiguous operands; an operand's std must have the operand's shape (one Measurand)."""
    try:
        shape = tuple(torch.broadcast_shapes(a.shape, b.shape))
    except RuntimeError:
        raise ValueError("Measurands are not broadcastable.")
    if len(shape) > nat.HM_MAX_DIMS:
        raise NotImplementedError(f"more than {nat.HM_MAX_DIMS} dimensions")
    if len(shape) == 0:
        shape = (1,)
    for v, s_ in ((a, sa), (b, sb)):
        if s_ is not None and s_.shape != v.shape:
            raise ValueError("Value and std shapes must match.")
    nd = len(shape)
    return shape, (C.c_int64 * nd)(*shape), (C.c_int64 * nd)(*_bcast_strides(a, shape)), (C.c_int64 * nd)(*_bcast_strides(b, shape))


def compute_difference(x, sx, y, sy, multiplier: float):
    """modules/measurand.py:620-655 -> (abs, abs_std, rel, rel_std). Operands of different shapes broadcast (hm_compute_difference_bcast)."""
    _require_cuda(x, "x")
    _require_cuda(y, "y")
    x, y = x.contiguous(), y.contiguous()
    if x.shape != y.shape:
        sx = None if sx is None else sx.contiguous()
        sy = None if sy is None else sy.contiguous()
        shape, sh, st1, st2 = _bcast_setup(x, sx, y, sy)
        with_std = sx is not None or sy is not None
        mk = lambda on: torch.empty(shape, dtype=_F64, device=x.device) if on else None      # noqa: E731
        ad, rd, ads, rds = mk(True), mk(True), mk(with_std), mk(with_std)
        with _on(x.device):
            nat.check(nat.lib.hm_compute_difference_bcast(x.data_ptr(), nat.ptr(sx), y.data_ptr(), nat.ptr(sy), float(multiplier), ad.data_ptr(),
                                                          nat.ptr(ads), rd.data_ptr(), nat.ptr(rds), len(shape), sh, st1, st2, _stream(x.device)),
                      "hm_compute_difference_bcast")
        return ad, ads, rd, rds
    sx = None if sx is None else sx.contiguous()
    sy = None if sy is None else sy.contiguous()
    with_std = sx is not None or sy is not None
    ad, rd = torch.empty_like(x), torch.empty_like(x)
    ads = torch.empty_like(x) if with_std else None
    rds = torch.empty_like(x) if with_std else None
    with _on(x.device):
        nat.check(nat.lib.hm_compute_difference(x.data_ptr(), nat.ptr(sx), y.data_ptr(), nat.ptr(sy), float(multiplier),
                                                ad.data_ptr(), nat.ptr(ads), rd.data_ptr(), nat.ptr(rds), x.numel(),
                                                _stream(x.device)), "hm_compute_difference")
    return ad, ads, rd, rds


def interpolate(x0, s0, x1, s1, y0: float, y1: float, y: float):
    """modules/measurand.py:657-681. Operands of different shapes broadcast (hm_interpolate_bcast)."""
    _require_cuda(x0, "x0")
    _require_cuda(x1, "x1")
    x0, x1 = x0.contiguous(), x1.contiguous()
    s0 = None if s0 is None else s0.contiguous()
    s1 = None if s1 is None else s1.contiguous()
    if x0.shape != x1.shape:
        shape, sh, st0, st1 = _bcast_setup(x0, s0, x1, s1)
        out = torch.empty(shape, dtype=_F64, device=x0.device)
        out_std = torch.empty(shape, dtype=_F64, device=x0.device) if (s0 is not None or s1 is not None) else None
        with _on(x0.device):
            nat.check(nat.lib.hm_interpolate_bcast(x0.data_ptr(), nat.ptr(s0), x1.data_ptr(), nat.ptr(s1), float(y0), float(y1), float(y),
                                                   out.data_ptr(), nat.ptr(out_std), len(shape), sh, st0, st1, _stream(x0.device)),
                      "hm_interpolate_bcast")
        return out, out_std
    out = torch.empty_like(x0)
    out_std = torch.empty_like(x0) if (s0 is not None or s1 is not None) else None
    with _on(x0.device):
        nat.check(nat.lib.hm_interpolate(x0.data_ptr(), nat.ptr(s0), x1.data_ptr(), nat.ptr(s1), float(y0), float(y1), float(y),
                                         out.data_ptr(), nat.ptr(out_std), x0.numel(), _stream(x0.device)), "hm_interpolate")
    return out, out_std


def channel_statistics(val: torch.Tensor, std: Optional[torch.Tensor]):
    """modules/measurand.py:318-350 over every axis but the last -> dict(mean, std, error) of (C,) device tensors."""
    _require_cuda(val, "val")
    val = val.contiguous()
    std = None if std is None else std.contiguous()
    Cc = val.shape[-1]
    out = torch.empty(3 * Cc, dtype=_F64, device=val.device)
    ws = torch.empty(nat.lib.hm_channel_statistics_workspace_bytes() // 8, dtype=_F64, device=val.device)
    with _on(val.device):
        nat.check(nat.lib.hm_channel_statistics(val.data_ptr(), nat.ptr(std), val.numel(), Cc, out.data_ptr(), ws.data_ptr(),
                                                _stream(val.device)), "hm_channel_statistics")
    return {"mean": out[:Cc], "std": out[Cc:2 * Cc], "error": out[2 * Cc:] if std is not None else None}


def axis_statistics(val: torch.Tensor, std: Optional[torch.Tensor], axis):
    """modules/measurand.py:318-350 for any `axis` (int or tuple, NumPy conventions) -> dict(mean, std, error) shaped like NumPy's
    result (the reduced axes removed). Adjacent reduced axes are reduced in place on the dense (outer, A, inner) view
    (hm_axis_statistics), two separate groups of them on the (outer, A1, mid, A2, inner) view (hm_axis_statistics2); only three or more
    separate groups are first brought together (the kept axes in order, then the reduced ones: a layout copy)."""
    _require_cuda(val, "val")
    nd = val.dim()
    axes = sorted({a % nd for a in ((axis,) if isinstance(axis, int) else tuple(axis))})
    if not axes or any(not -nd <= a < nd for a in ((axis,) if isinstance(axis, int) else tuple(axis))):
        raise ValueError(f"axis {axis} is out of bounds for an array of dimension {nd}")
    if std is not None:
        _require_cuda(std, "std")
        if std.shape != val.shape:
            raise ValueError("Value and std shapes must match.")
        std = std.to(_F64)
    kept = [d for d in range(nd) if d not in axes]
    out_shape = tuple(val.shape[d] for d in kept)
    groups = [[axes[0]]]                                       # runs of adjacent reduced axes
    for a_ in axes[1:]:
        if a_ == groups[-1][-1] + 1:
            groups[-1].append(a_)
        else:
            groups.append([a_])
    if len(groups) > 2:                                        # three or more separate groups: bring them together (a layout copy)
        perm = kept + axes
        val = val.permute(perm)
        std = None if std is None else std.permute(perm)
        groups = [list(range(len(kept), nd))]
    val = val.contiguous()
    std = None if std is None else std.contiguous()
    sh = list(val.shape)
    prod = lambda lo, hi: int(np.prod(sh[lo:hi], dtype=np.int64))   # noqa: E731
    if prod(0, len(sh)) == 0:
        raise ValueError("statistics of an empty array")
    dev = val.device
    mean = torch.empty(out_shape, dtype=_F64, device=dev)
    sd = torch.empty(out_shape, dtype=_F64, device=dev)
    err = torch.empty(out_shape, dtype=_F64, device=dev) if std is not None else None
    if len(groups) == 1:
        g = groups[0]
        outer, A, inner = prod(0, g[0]), prod(g[0], g[-1] + 1), prod(g[-1] + 1, len(sh))
        ws_b = int(nat.lib.hm_axis_statistics_workspace_bytes(outer, A, inner))
        ws = torch.empty(max(1, ws_b // 8), dtype=_F64, device=dev)
        with _on(dev):
            nat.check(nat.lib.hm_axis_statistics(val.data_ptr(), nat.ptr(std), outer, A, inner, mean.data_ptr(), sd.data_ptr(), nat.ptr(err),
                                                 ws.data_ptr(), _stream(dev)), "hm_axis_statistics")
    else:                                                      # two groups with kept axes between them: reduced in place, no layout copy
        g1, g2 = groups
        dims = (prod(0, g1[0]), prod(g1[0], g1[-1] + 1), prod(g1[-1] + 1, g2[0]), prod(g2[0], g2[-1] + 1), prod(g2[-1] + 1, len(sh)))
        ws_b = int(nat.lib.hm_axis_statistics2_workspace_bytes(*dims))
        ws = torch.empty(max(1, ws_b // 8), dtype=_F64, device=dev)
        with _on(dev):
            nat.check(nat.lib.hm_axis_statistics2(val.data_ptr(), nat.ptr(std), *dims, mean.data_ptr(), sd.data_ptr(), nat.ptr(err),
                                                  ws.data_ptr(), _stream(dev)), "hm_axis_statistics2")
    return {"mean": mean, "std": sd, "error": err}


def pair_statistics(x, sx, y, sy, multiplier: float):
    """ExposurePair.compute_difference + compute_stats(axis=(0,1)) in one fused reduction (no difference images):
    -> (absolute_stats, relative_stats), each dict(mean, std, error) of (C,) device tensors."""
    _require_cuda(x, "x")
    _require_cuda(y, "y")
    if x.shape != y.shape:
        raise ValueError("pair statistics need frames of equal shape")
    x, y = x.contiguous(), y.contiguous()
    sx = None if sx is None else sx.contiguous()
    sy = None if sy is None else sy.contiguous()
    Cc = x.shape[-1]
    out = torch.empty(6 * Cc, dtype=_F64, device=x.device)
    ws = torch.empty(nat.lib.hm_pair_statistics_workspace_bytes() // 8, dtype=_F64, device=x.device)
    with _on(x.device):
        nat.check(nat.lib.hm_pair_statistics(x.data_ptr(), nat.ptr(sx), y.data_ptr(), nat.ptr(sy), float(multiplier), x.numel(), Cc,
                                             out.data_ptr(), ws.data_ptr(), _stream(x.device)), "hm_pair_statistics")
    w = sx is not None or sy is not None
    mk = lambda o: {"mean": out[o:o + Cc], "std": out[o + Cc:o + 2 * Cc], "error": out[o + 2 * Cc:o + 3 * Cc] if w else None}   # noqa: E731
    return mk(0), mk(3 * Cc)


def pairs_statistics(vals: Sequence[torch.Tensor], stds: Optional[Sequence[torch.Tensor]], pairs: Sequence[tuple], to_host: bool = False,
                     thresholds: Optional[tuple] = None):
    """Statistics of the absolute and relative difference of EVERY exposure pair of a stack in one fused launch
    (hm_pairs_statistics; modules/exposure_series.py:421-446). `pairs` = [(i, j, multiplier), ...] with i the short and j the
    long exposure. Returns one (abs_stats, rel_stats) tuple of dicts per pair, as pair_statistics() does.
    thresholds = (lower, upper), C numbers each: apply_thresholds(lower, upper) is applied to every frame (and std) IN PLACE first,
    fused into the launch's loads - the frames must then be the contiguous float64 tensors the caller keeps."""
    n = len(vals)
    for i, v in enumerate(vals):
        _require_cuda(v, f"vals[{i}]")
        if v.shape != vals[0].shape or v.dtype != _F64:
            raise ValueError("pairs_statistics needs float64 frames of one shape")
    dev = vals[0].device
    if thresholds is not None and (any(not v.is_contiguous() for v in vals) or
                                   (stds is not None and any(s is None or s.dtype != _F64 or not s.is_contiguous() for s in stds))):
        raise ValueError("in-place thresholds need contiguous float64 frames (and stds)")
    vals = [v.contiguous() for v in vals]
    if stds is not None:
        if len(stds) != n or any(s is None for s in stds):
            raise ValueError("one std frame per value frame (or none at all)")
        stds = [s.to(_F64).contiguous() for s in stds]
    Cc = vals[0].shape[-1]
    lo = hi = None
    if thresholds is not None:
        if len(thresholds[0]) != Cc or len(thresholds[1]) != Cc:
            raise ValueError("The length of 'lower' and 'upper' must match the size of the independent axis.")
        lo = (C.c_double * Cc)(*[float(x) for x in thresholds[0]])
        hi = (C.c_double * Cc)(*[float(x) for x in thresholds[1]])
    P = len(pairs)
    vp = _ptr_array(vals)
    sp = None if stds is None else _ptr_array(stds)
    pi = (C.c_int32 * P)(*[int(p[0]) for p in pairs])
    pj = (C.c_int32 * P)(*[int(p[1]) for p in pairs])
    pm = (C.c_double * P)(*[float(p[2]) for p in pairs])
    out = torch.empty(P * 6 * Cc, dtype=_F64, device=dev)
    ws = torch.empty(max(1, nat.lib.hm_pairs_statistics_workspace_bytes(P) // 8), dtype=_F64, device=dev)
    with _on(dev):
        nat.check(nat.lib.hm_pairs_statistics(C.cast(vp, C.POINTER(C.c_void_p)), None if sp is None else C.cast(sp, C.POINTER(C.c_void_p)), n,
                                              pi, pj, pm, P, vals[0].numel(), Cc, lo, hi, out.data_ptr(), ws.data_ptr(), _stream(dev)),
                  "hm_pairs_statistics")
    w = stds is not None
    if to_host:
        out = out.cpu()                                   # one device-to-host copy for all pairs (P * 6C numbers)
    res = []
    for p in range(P):
        b = p * 6 * Cc
        mk = lambda o: {"mean": out[o:o + Cc], "std": out[o + Cc:o + 2 * Cc], "error": out[o + 2 * Cc:o + 3 * Cc] if w else None}   # noqa: E731
        res.append((mk(b), mk(b + 3 * Cc)))
    return res


def channel_histogram(val: torch.Tensor, std: Optional[torch.Tensor], bins: int, included_range, channels: Sequence[int]):
    """modules/measurand.py:430-469 -> {c: (hist ndarray, bin_edges ndarray)} like np.histogram. A range of None is
    evaluated per channel from the counted values (np.histogram's default), with one extra reduction."""
    _require_cuda(val, "val")
    val = val.contiguous()
    std = None if std is None else std.contiguous()
    Cc = val.shape[-1]
    ws = torch.empty(max(1, nat.lib.hm_histogram_workspace_bytes(int(bins), Cc) // 8), dtype=_F64, device=val.device)
    out = {}
    with _on(val.device):
        if included_range is None:
            mm = torch.empty(2 * Cc, dtype=_F64, device=val.device)
            nat.check(nat.lib.hm_channel_minmax(val.data_ptr(), nat.ptr(std), val.numel(), Cc, mm.data_ptr(), ws.data_ptr(),
                                                _stream(val.device)), "hm_channel_minmax")
            mmh = mm.cpu().numpy().reshape(Cc, 2)
        groups = {}
        for c in channels:
            lo, hi = (float(mmh[c, 0]), float(mmh[c, 1])) if included_range is None else (float(included_range[0]), float(included_range[1]))
            if lo == hi:                                   # np.histogram widens an empty range by +-0.5
                lo, hi = lo - 0.5, hi + 0.5
            groups.setdefault((lo, hi), []).append(c)
        for (lo, hi), cs in groups.items():
            edges = np.linspace(lo, hi, int(bins) + 1)
            edges_d = torch.as_tensor(edges, device=val.device)
            res = torch.empty(Cc * int(bins), dtype=_F64, device=val.device)
            mask = sum(1 << c for c in cs)
            nat.check(nat.lib.hm_channel_histogram(val.data_ptr(), nat.ptr(std), val.numel(), Cc, mask, edges_d.data_ptr(), int(bins), lo, hi,
                                                   res.data_ptr(), ws.data_ptr(), _stream(val.device)), "hm_channel_histogram")
            r = res.cpu().numpy().reshape(Cc, int(bins))
            for c in cs:
                out[c] = (r[c] if std is not None else r[c].astype(np.int64), edges)
    return out


# ------------------------------------------------------------------------------------------------
# upstream producer: Welford mean / M2 over video frames (modules/video_processing.py:161-219)
# ------------------------------------------------------------------------------------------------
def welford_update(frames: Sequence[torch.Tensor], count_before: int, mean: torch.Tensor, m2: Optional[torch.Tensor],
                   icrf=None) -> int:
    """Fold `frames` (uint8 (H, W, C) device tensors, in order) into the running float64 mean / m2 in place
    (video_processing.py:199-208). Returns the new frame count. Frames are folded HM_MAX_FRAMES per launch."""
    _require_cuda(mean, "mean")
    if mean.dtype != _F64 or not mean.is_contiguous():
        raise TypeError("mean must be a contiguous float64 tensor")
    if m2 is not None and (m2.dtype != _F64 or not m2.is_contiguous() or m2.shape != mean.shape):
        raise TypeError("m2 must be a contiguous float64 tensor shaped like mean")
    dev = mean.device
    Cc = mean.shape[-1]
    lut = None if icrf is None else _dev_f64(icrf, dev)
    if lut is not None and tuple(lut.shape) != (BITS, Cc):
        raise ValueError(f"ICRF must be shaped ({BITS}, {Cc})")
    count = int(count_before)
    frames = list(frames)
    for f in frames:
        _require_cuda(f, "frame")
        if f.dtype != torch.uint8 or f.shape != mean.shape:
            raise ValueError("every frame must be a uint8 tensor shaped like mean")
    with _on(dev):
        for k0 in range(0, len(frames), nat.HM_MAX_FRAMES):
            batch = [f.contiguous() for f in frames[k0:k0 + nat.HM_MAX_FRAMES]]
            nat.check(nat.lib.hm_welford_update(_ptr_array(batch), len(batch), count, nat.ptr(lut), mean.data_ptr(), nat.ptr(m2),
                                                mean.numel(), Cc, _stream(dev)), "hm_welford_update")
            count += len(batch)
    return count


def welford_finalize(mean: torch.Tensor, m2: Optional[torch.Tensor], count: int):
    """-> (mean frame uint8, std frame uint8 or None), video_processing.py:210-215."""
    _require_cuda(mean, "mean")
    out_mean = torch.empty(mean.shape, dtype=torch.uint8, device=mean.device)
    out_std = None if m2 is None else torch.empty(mean.shape, dtype=torch.uint8, device=mean.device)
    with _on(mean.device):
        nat.check(nat.lib.hm_welford_finalize(mean.data_ptr(), nat.ptr(m2), int(count), out_mean.data_ptr(), nat.ptr(out_std),
                                              mean.numel(), _stream(mean.device)), "hm_welford_finalize")
    return out_mean, out_std


# ------------------------------------------------------------------------------------------------
# ICRF-calibration energy function (modules/ICRF_calibration_exposure.py:66-201), batched over candidates
# ------------------------------------------------------------------------------------------------
def linearity_energy(dn_stack: torch.Tensor, std_stack: Optional[torch.Tensor], exposures: Sequence[float], icrf_batch,
                     lower: int, upper: int, valid=None, use_relative: bool = True, return_pairs: bool = False):
    """Energy of every candidate ICRF row of `icrf_batch` ((B, 256) float64) on one channel's (X, Y, N) uint8 stack.
    -> energies (B,) float64 device tensor [, pair results (B, N(N-1)/2)]."""
    _require_cuda(dn_stack, "image_value_stack")
    if dn_stack.dtype != torch.uint8:
        raise TypeError("image_value_stack must be uint8 digital numbers")
    if dn_stack.dim() != 3:
        raise ValueError("image_stack must be a 3D array with shape (X, Y, N).")             # ICRF_calibration_exposure.py:82-83
    dev = dn_stack.device
    N = dn_stack.shape[2]
    t = np.asarray(exposures, dtype=np.float64)
    if t.ndim != 1 or t.size != N:
        raise ValueError("exposure_values must be a 1D array matching the third dimension of image_stack.")   # :85-86
    dn_stack = dn_stack.contiguous()
    if std_stack is not None:
        _require_cuda(std_stack, "image_std_stack")
        if std_stack.shape != dn_stack.shape or std_stack.dtype != _F64:
            raise ValueError("image_std_stack must be float64 and shaped like image_value_stack")
        std_stack = std_stack.contiguous()
    lut = _dev_f64(icrf_batch, dev)
    if lut.dim() == 1:
        lut = lut[None]
    if lut.shape[1] != BITS:
        raise ValueError(f"candidate ICRFs must have {BITS} entries")
    lut = lut.contiguous()
    B = lut.shape[0]
    vd = None if valid is None else torch.as_tensor(np.asarray(valid, dtype=np.uint8), device=dev)
    P = dn_stack.shape[0] * dn_stack.shape[1]
    pairs = N * (N - 1) // 2
    energy = torch.empty(B, dtype=_F64, device=dev)
    out_pairs = torch.empty((B, pairs), dtype=_F64, device=dev) if return_pairs else None
    ws = torch.empty(max(1, nat.lib.hm_linearity_energy_workspace_bytes(P, N, B) // 8), dtype=_F64, device=dev)
    with _on(dev):
        nat.check(nat.lib.hm_linearity_energy(dn_stack.data_ptr(), nat.ptr(std_stack), (C.c_double * N)(*t.tolist()), lut.data_ptr(),
                                              nat.ptr(vd), B, int(lower), int(upper), int(bool(use_relative)), P, N,
                                              nat.ptr(out_pairs), energy.data_ptr(), ws.data_ptr(), _stream(dev)),
                  "hm_linearity_energy")
    return (energy, out_pairs) if return_pairs else energy
