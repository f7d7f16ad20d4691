"""Host-side helpers of modules/general_functions.py under the reference's module name (`import general_functions as gf` users): small
NumPy utilities of the callers around the hot path - none of them touches an image-sized array on the device. The ones this package already
defines where it uses them are re-exported (one definition each); the rest are restated here from the reference's documented behaviour and
pinned to outputs the reference itself produced (tests/golden/helpers.npz, tests/test_general_functions.py).

`use_cupy` arguments are accepted for signature compatibility: ICRF tables and text data are HOST arrays in this package (the engine places
tables on the device of the images they are used with).
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Optional

import numpy as np

from . import settings as gs
from .exposure_series import map_linearity_limits as _map_linearity_limits, read_ICRF_file as _read_icrf_file
from .measurand import is_broadcastable                                      # modules/general_functions.py:14-24
from .video_processing import frames_from_capture

__all__ = ["is_broadcastable", "choose_evenly_spaced_points", "predict_output_shape", "interpolate_data", "map_linearity_limits",
           "weighted_avg_and_std", "nanaverage", "weighted_percentile", "video_frame_generator", "read_ICRF_file", "read_txt_to_array"]


def _host(a):
    """NumPy view of a NumPy array or torch tensor (these helpers compute on the host)."""
    if a is None:
        return None
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def choose_evenly_spaced_points(array, step_x: int, step_y: Optional[int] = None):
    """modules/general_functions.py:27-46: every step_x-th row and step_y-th column (step_y defaults to step_x); a view, as in NumPy."""
    step_y = step_x if step_y is None else step_y
    return array[::step_x, ::step_y, ...]


def predict_output_shape(input_shape, step_x: int, step_y: Optional[int] = None):
    """:49-69: the (rows, columns) choose_evenly_spaced_points returns for an input of `input_shape` - ceil(rows / step_x), ceil(cols / step_y)."""
    step_y = step_x if step_y is None else step_y
    rows, cols = input_shape
    return -(-rows // step_x), -(-cols // step_y)


def interpolate_data(clean_data_arr: np.ndarray):
    """:72-94: resample every row from BITS to DATAPOINTS samples on [0, 1] by linear interpolation (unchanged when the two agree)."""
    if gs.BITS == gs.DATAPOINTS:
        return clean_data_arr
    x_old = np.linspace(0, 1, num=gs.BITS)
    x_new = np.linspace(0, 1, num=gs.DATAPOINTS)
    out = np.zeros((gs.BITS, gs.DATAPOINTS), dtype=float)
    for i in range(gs.BITS):
        out[i, :] = np.interp(x_new, x_old, clean_data_arr[i, :])
    return out


def map_linearity_limits(lower_limit: Optional[int], upper_limit: Optional[int], ICRF):
    """:97-128 -> (lower, upper) float64 arrays of NUM_OF_CHS entries (exposure_series.map_linearity_limits returns the same numbers as lists)."""
    lo, hi = _map_linearity_limits(lower_limit, upper_limit, ICRF)
    return np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)


def weighted_avg_and_std(values, weights):
    """:131-146: np.average with weights (None = plain), and the square root of the equally weighted average of (values - average)**2."""
    v, w = _host(values), _host(weights)
    average = np.average(v, weights=w)
    variance = np.average((v - average) ** 2, weights=w)
    return average, math.sqrt(variance)


def nanaverage(values, weights, axis):
    """:149-175: weighted average along `axis` over the positions where neither the value nor the weight is NaN; NaN where no weight is left."""
    v, w = _host(values), _host(weights)
    valid = ~np.isnan(v) & ~np.isnan(w)
    with np.errstate(invalid="ignore", divide="ignore"):
        num = np.nansum(v * w * valid, axis=axis)
        den = np.nansum(valid * w, axis=axis)
        res = np.asarray(num / den)
    res[den == 0] = np.nan
    return res


def weighted_percentile(values, percentiles=None, weights=None):
    """:178-223: percentiles (default 75 and 25) of `values` with integer-like weights acting as repeat counts: sort, cumulative weights,
    position p = q (sum(w) - 1), the sorted values at the two bounding positions blended by the fractional part of p."""
    v = _host(values)
    q = np.array([75, 25]) if percentiles is None else np.asarray(_host(percentiles))
    q = q / 100.0
    w = np.ones(v.size) if weights is None else _host(weights)
    order = np.argsort(v)
    v_sorted, w_sorted = v[order], w[order]
    ecdf = np.cumsum(w_sorted)
    p = q * (w.sum() - 1)
    i_lo = np.searchsorted(ecdf, p, side="right")
    i_hi = np.searchsorted(ecdf, p + 1, side="right")
    i_hi[i_hi > ecdf.size - 1] = ecdf.size - 1
    frac = p - np.floor(p)
    return np.take(v_sorted, i_lo) * (1.0 - frac) + np.take(v_sorted, i_hi) * frac


def video_frame_generator(video_path):
    """:226-251: frames of the video at `video_path`, then None. Needs OpenCV (`cv2.VideoCapture`), which this package does not depend on:
    with a capture-like object in hand use video_processing.frames_from_capture(capture) instead."""
    try:
        import cv2 as cv
    except ImportError as e:
        raise ImportError("video_frame_generator opens the file with cv2.VideoCapture; OpenCV is not installed. "
                          "video_processing.frames_from_capture(capture) takes any opened capture-like object.") from e
    capture = cv.VideoCapture(str(video_path))
    if not capture.isOpened():
        raise ValueError(f"Unable to open video file at {video_path}")
    yield from frames_from_capture(capture)
    yield None


def read_ICRF_file(file_path, return_derivative: Optional[bool] = True, use_cupy: Optional[bool] = False):
    """:254-277 -> (ICRF, ICRF_diff | None), the derivative with dx = 2 / (BITS - 1) (:270). (As written the reference returns the ICRF
    a second time in place of the derivative - `ICRF_diff = cast_to_array(ICRF, ...)`, :275; the derivative is what its callers expect.)"""
    return _read_icrf_file(file_path, bool(return_derivative))


def read_txt_to_array(file_name: str, path: Optional[str] = None, use_cupy: Optional[bool] = True):
    """:280-302: np.loadtxt(<path or settings.DATA_PATH> / file_name) as float64."""
    base = getattr(gs, "DATA_PATH", None) if path is None else Path(path)
    if base is None:
        raise ValueError("read_txt_to_array needs `path` or settings.configure(DATA_PATH=...)")
    return np.loadtxt(Path(base).joinpath(file_name), dtype=float)
