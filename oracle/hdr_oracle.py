"""CPU oracle for the HDR-merge / linearization hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the reference algorithm (samivout/camera_linearity,
snapshot 2025-03-21). It is the checker for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it. Nothing under
`camera_linearity_amd/` imports, calls or links anything from `oracle/`; the product path has no CPU
fallback and raises when the HIP library is missing.

Pinning: every function here is checked (tests/test_oracle_golden.py) against the vectors in
tests/golden/*.npz, which were produced by tests/golden/make_golden.py RUNNING the reference's own
classes in the build container (`apply_gaussian_weight`, per-channel `_linearize_single`,
`__add__`/`__pow__`, ... - SURVEY.md section 8c). The reference's merge loop
(`process_HDR_image`) does not execute at HEAD; the merge below evaluates the
formulas of modules/exposure_series.py:388-389,394 verbatim with deviations A-J of that table and
nothing else. The hot-pixel filter and the flat-field step are unpinned by any reference TEST
 - they are pinned by the generated vectors only.

Every function cites the reference file:line it follows. Operation order is kept identical to
the reference so that results agree to the last few ulps.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

BITS = 256       # modules/global_settings.py:36  (BIT_DEPTH = 8)
MAX_DN = 255     # modules/global_settings.py:37


# --------------------------------------------------------------------------------------------
# inputs' conventions
# --------------------------------------------------------------------------------------------
def unit_from_u8(dn: np.ndarray) -> np.ndarray:
    """modules/image_set.py:223 - `cv.imread(...).astype(np.float64) / gs.MAX_DN`."""
    return dn.astype(np.float64) / MAX_DN


def icrf_derivative(icrf: np.ndarray, bits: int = BITS) -> np.ndarray:
    """modules/general_functions.py:268-272 (intended: the gradient is what is returned) and
    tests/unit/test_measurand.py:21 - `np.gradient(ICRF[:, c], 2 / (BITS - 1))` per channel."""
    icrf = np.asarray(icrf, dtype=np.float64)
    dx = 2 / (bits - 1)
    if icrf.ndim == 1:
        return np.gradient(icrf, dx)
    out = np.zeros_like(icrf)
    for c in range(icrf.shape[1]):
        out[:, c] = np.gradient(icrf[:, c], dx)
    return out


# --------------------------------------------------------------------------------------------
# row 3: apply_gaussian_weight
# --------------------------------------------------------------------------------------------
def gaussian_weight(v: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """modules/measurand.py:615-616 - note `np.e ** x`, not `np.exp(x)`."""
    y = np.e ** (-30 * (v - 0.5) ** 2)
    dydx = -2 * 30 * (v - 0.5) * y
    return y, dydx


def gaussian_weight_lut(bits: int = BITS) -> Tuple[np.ndarray, np.ndarray]:
    """w, dw evaluated on the DN grid v = k / MAX_DN (exactly the values an 8-bit frame takes
    after modules/image_set.py:223)."""
    v = np.arange(bits, dtype=np.uint8 if bits <= 256 else np.int64).astype(np.float64) / (bits - 1)
    return gaussian_weight(v)


# --------------------------------------------------------------------------------------------
# row 4: linearize
# --------------------------------------------------------------------------------------------
def lut_index(val: np.ndarray) -> np.ndarray:
    """modules/measurand.py:502-505 / :530-533 - integer input is used as is, float input is
    `around(val * MAX_DN).astype(uint8)` (round half to even, then C cast to uint8)."""
    if np.issubdtype(val.dtype, np.integer):
        return val.copy()
    return np.around(val * MAX_DN).astype(np.dtype("uint8"))


def linearize(val: np.ndarray, std: Optional[np.ndarray], icrf: np.ndarray,
              icrf_diff: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Optional[np.ndarray], np.ndarray]:
    """modules/measurand.py:471-541 with the intended multi-channel semantics
    out[..., c] = ICRF[idx[..., c], c] (deviation A/B; the semantics of tests/unit/test_measurand.py:463-467
    and modules/video_processing.py:201). A 1-D ICRF applies to every element (`_linearize_single`).
    Returns (val, std or None, idx)."""
    idx = lut_index(val)
    icrf = np.asarray(icrf)
    if icrf.ndim == 1:
        out = icrf[idx]
        dtab = None if icrf_diff is None else np.asarray(icrf_diff)[idx]
    else:
        ch = np.arange(val.shape[-1])
        out = icrf[idx, ch]
        dtab = None if icrf_diff is None else np.asarray(icrf_diff)[idx, ch]
    if std is None or icrf_diff is None:           # measurand.py:498-500
        return out, None, idx
    return out, dtab * std, idx                    # measurand.py:512 / :539


# --------------------------------------------------------------------------------------------
# row 8: dark frame selection + hot pixel filter
# --------------------------------------------------------------------------------------------
def select_dark(target_exposure: float, dark_exposures: Sequence[float], dark_threshold: float
                ) -> Tuple[int, float]:
    """modules/image_set.py:171-198. Returns (index into the dark list, scale) or (-1, 0.0).
    The list is walked in order; an exact-exposure dark wins, otherwise, as soon as both a shorter
    and a longer dark have been seen, the most recently seen longer one is scaled by
    target/dark exposure (scale_to_exposure, image_set.py:245-262 with deviation I)."""
    if not (target_exposure >= dark_threshold):
        return -1, 0.0
    lesser = greater = False
    greater_index = 0
    for i, de in enumerate(dark_exposures):
        if de < target_exposure:
            lesser = True
        if de > target_exposure:
            greater = True
            greater_index = i
        if target_exposure == de:
            return i, 1.0
        if lesser and greater:
            return greater_index, target_exposure / dark_exposures[greater_index]
    return -1, 0.0


def median_filter_reflect(x: np.ndarray, k: int) -> np.ndarray:
    """k x k spatial median over axes (0, 1) with scipy.ndimage 'reflect' boundary (d c b a | a b c d),
    the call of modules/measurand.py:546-547. Written out with NumPy (np.pad 'symmetric' is the same
    boundary rule) so the oracle has no SciPy dependency on the GPU box; checked against SciPy in tests."""
    r = k // 2
    pad = [(r, r), (r, r)] + [(0, 0)] * (x.ndim - 2)
    xp = np.pad(x, pad, mode="symmetric")
    h, w = x.shape[:2]
    stack = np.stack([xp[dy:dy + h, dx:dx + w] for dy in range(k) for dx in range(k)], axis=0)
    stack = np.sort(stack, axis=0)
    return stack[(k * k) // 2]


def hot_pixel_filter(x: np.ndarray, dark_val: np.ndarray, thr: float, k: int) -> np.ndarray:
    """Intended semantics of modules/measurand.py:543-557 (deviation F): pixels where the dark map
    exceeds the threshold are replaced by the k x k median of the frame, the others are kept."""
    return np.where(dark_val > thr, median_filter_reflect(x, k), x)


# --------------------------------------------------------------------------------------------
# row 9: flat field
# --------------------------------------------------------------------------------------------
def flat_roi_bounds(size_x: int, size_y: int, p: float) -> Tuple[int, int, int, int]:
    """modules/measurand.py:570-576 with the integer ROI index of deviation H."""
    dx = math.floor(size_x * p)
    dy = math.floor(size_y * p)
    i = (math.floor(1 / p) - 1) // 2
    return i * dx, (i + 1) * dx, i * dy, (i + 1) * dy


def flat_roi_mean(flat: np.ndarray, size_x: int, size_y: int, p: float) -> np.ndarray:
    """modules/measurand.py:579 - per-channel mean over the centred ROI."""
    x0, x1, y0, y1 = flat_roi_bounds(size_x, size_y, p)
    return np.mean(flat[x0:x1, y0:y1, :], axis=(0, 1))


def normalize_by_map(val, std, fval, fstd, ff_mean, ff_std_mean):
    """modules/measurand.py:585-604, operation for operation."""
    u_acq = (std ** 2) / (fval ** 2)
    u_acq *= ff_mean ** 2
    u_ff = (val ** 2) / (fval ** 4)
    u_ff *= fstd ** 2
    u_ff *= ff_mean ** 2
    u_ffm = (val ** 2) / (fval ** 2)
    u_ffm *= ff_std_mean ** 2
    ret_std = np.sqrt(u_acq + u_ff + u_ffm)
    ret_val = (val / fval) * ff_mean
    return ret_val, ret_std


# --------------------------------------------------------------------------------------------
# rows 5-7: the merge
# --------------------------------------------------------------------------------------------
def sum_of_weights(frames_val: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """modules/exposure_series.py:331-345 (deviation D: plain arrays). S starts from zeros and the
    weights are added in stack order; S2 = S ** 2."""
    S = np.zeros_like(frames_val[0], dtype=np.float64)
    for v in frames_val:
        S = S + gaussian_weight(v)[0]
    return S, S ** 2


def merge(frames: Sequence[np.ndarray], exposures: Sequence[float], icrf: np.ndarray,
          icrf_diff: Optional[np.ndarray] = None, stds: Optional[Sequence[np.ndarray]] = None,
          darks: Optional[Sequence[Optional[np.ndarray]]] = None, dark_threshold: float = 0.0,
          median_k: int = 3, flat: Optional[np.ndarray] = None, flat_std: Optional[np.ndarray] = None,
          ff_mean: Optional[np.ndarray] = None, ff_std_mean: Optional[np.ndarray] = None):
    """modules/exposure_series.py:317-419 on in-memory stacks.

    frames  : N arrays (H, W, C), uint8 DNs or float64 values in [0, 1]; ascending exposure.
    stds    : N float64 arrays or None (val-only mode = line :388 alone).
    darks   : per frame, the dark map as float64 *values* (DN/255 x scale) or None (no filtering).
    flat    : float64 flat-field values; with flat_std, ff_mean, ff_std_mean applies :559-604.
    Returns dict(val, std, S, idx[N]).
    """
    vals, sds = [], []
    for i, f in enumerate(frames):
        v = unit_from_u8(f) if np.issubdtype(f.dtype, np.integer) else f
        s = None if stds is None else stds[i]
        if darks is not None and darks[i] is not None:          # exposure_series.py:337-339, 379-380
            v = hot_pixel_filter(v, darks[i], dark_threshold, median_k)
            if s is not None:
                s = hot_pixel_filter(s, darks[i], dark_threshold, median_k)
        vals.append(v)
        sds.append(s)
    S, S2 = sum_of_weights(vals)                                # exposure_series.py:411
    hdr_val = np.zeros_like(vals[0])
    hdr_std = None if stds is None else np.zeros_like(vals[0])
    idxs = []
    for i, v in enumerate(vals):
        w, dw = gaussian_weight(v)                              # :382 (pre-linearization value)
        g, dg, idx = linearize(v, sds[i], icrf, icrf_diff)      # :383
        idxs.append(idx)
        t = exposures[i]                                        # :386
        hdr_val += (w * g) / (S * t)                            # :388
        if hdr_std is not None:
            hdr_std += (((dw * g + w * dg) / S - (dw * w * g) / S2) * dg / t) ** 2   # :389
    if hdr_std is not None:
        hdr_std = hdr_std ** (1 / 2)                            # :394
    out = dict(val=hdr_val, std=hdr_std, S=S, idx=np.stack(idxs))
    if flat is not None:                                        # :415-417
        fv, fs = normalize_by_map(hdr_val, hdr_std, flat, flat_std, ff_mean, ff_std_mean)
        out["val_ff"], out["std_ff"] = fv, fs
    return out


# --------------------------------------------------------------------------------------------
# row 10: Measurand operators (first-order propagation)
# --------------------------------------------------------------------------------------------
def _z(s, x):
    return np.zeros_like(x) if s is None else s


def op_add(x1, s1, x2, s2):
    """modules/measurand.py:106-128."""
    r = x1 + x2
    if s1 is None and s2 is None:
        return r, None
    return r, np.sqrt((_z(s1, x1) ** 2) + (_z(s2, x2) ** 2))


def op_sub(x1, s1, x2, s2):
    """modules/measurand.py:130-150."""
    r = x1 - x2
    if s1 is None and s2 is None:
        return r, None
    return r, np.sqrt((_z(s1, x1) ** 2) + (_z(s2, x2) ** 2))


def op_mul(x1, s1, x2, s2):
    """modules/measurand.py:190-211."""
    r = x1 * x2
    if s1 is None and s2 is None:
        return r, None
    return r, np.sqrt((x1 * _z(s2, x2)) ** 2 + (x2 * _z(s1, x1)) ** 2)


def op_div(x1, s1, x2, s2):
    """modules/measurand.py:165-188."""
    r = x1 / x2
    if s1 is None and s2 is None:
        return r, None
    u1 = _z(s1, x1) / x2
    u2 = (x1 * _z(s2, x2)) / (x2 ** 2)
    return r, np.sqrt(u1 ** 2 + u2 ** 2)


def op_pow(x1, s1, x2, s2):
    """modules/measurand.py:217-241."""
    r = x1 ** x2
    if s1 is None and s2 is None:
        return r, None
    u1 = (x2 * x1 ** (x2 - 1))
    u2 = (np.log(x1) * x1 ** x2)
    return r, np.sqrt((u1 * _z(s1, x1)) ** 2 + (u2 * _z(s2, x2)) ** 2)


def op_neg(x, s):
    """modules/measurand.py:152-163."""
    return np.negative(x), (None if s is None else s.copy())


def op_log_e(x, s):
    """modules/measurand.py:243-260 (std = std / log(x), as written)."""
    return np.log(x), (None if s is None else s / np.log(x))


def op_log_10(x, s):
    """modules/measurand.py:262-279."""
    return np.log10(x), (None if s is None else s / (x * (np.log(5) + np.log(2))))


def compute_difference(x, sx, y, sy, multiplier):
    """modules/measurand.py:620-655. Returns (abs, abs_std, rel, rel_std)."""
    scale = multiplier * y
    abs_diff = x - scale
    rel_diff = abs_diff / scale
    if sx is None and sy is None:
        return abs_diff, None, rel_diff, None
    x_std = 0 if sx is None else sx
    y_std = 0 if sy is None else sy
    abs_std = np.sqrt(x_std ** 2 + (multiplier * y_std) ** 2)
    rel_std = np.sqrt((x_std / (multiplier * y)) ** 2 + ((y_std * x) / (multiplier * y ** 2)) ** 2)
    return abs_diff, abs_std, rel_diff, rel_std


def interpolate(x0, s0, x1, s1, y0, y1, y):
    """modules/measurand.py:657-681 (std formula as written: std, not std**2, under the root)."""
    res = (x0 * (y1 - y) + x1 * (y - y0)) / (y1 - y0)
    if s0 is None and s1 is None:
        return res, None
    a = 0 if s0 is None else s0
    b = 0 if s1 is None else s1
    return res, np.sqrt(a * ((y1 - y) / (y1 - y0)) ** 2 + b * ((y - y0) / (y1 - y0)) ** 2)


def dimension_statistics(val, std, axis):
    """modules/measurand.py:318-350."""
    if std is None:
        return dict(mean=np.nanmean(val, axis=axis), std=np.nanstd(val, axis=axis), error=None)
    weights = 1 / std
    sw = np.nansum(weights, axis=axis)
    mean = np.nansum(val * weights, axis=axis) / sw
    sd = np.sqrt(np.nansum(weights * (val - mean) ** 2, axis=axis) / sw)
    return dict(mean=mean, std=sd, error=np.nanmean(std, axis=axis))


def apply_thresholds(val, std, lower, upper):
    """modules/measurand.py:375-428 (returns new arrays instead of mutating)."""
    n = val.shape[-1]
    lower = [None] * n if lower is None else lower
    upper = [None] * n if upper is None else upper
    if len(lower) != n or len(upper) != n:
        raise ValueError("The length of 'lower' and 'upper' must match the size of the independent axis.")
    lo = np.array([l if l is not None else -np.inf for l in lower], dtype=val.dtype)
    hi = np.array([u if u is not None else np.inf for u in upper], dtype=val.dtype)
    mask = (val < lo) | (val > hi)
    v = val.copy()
    v[mask] = np.nan
    s = None
    if std is not None:
        s = std.copy()
        s[mask] = np.nan
    return v, s


# --------------------------------------------------------------------------------------------
# upstream producer: Welford mean / std frames (modules/video_processing.py:161-219)
# --------------------------------------------------------------------------------------------
def welford_state(frames, icrf=None, use_std=True, mean=None, m2=None, count=0):
    """Running state after folding `frames` (uint8 (H, W, C) each): video_processing.py:199-208.
    Deviation K: `if ICRF:` (:200) raises for an ndarray at HEAD; the intended test is `ICRF is not None`."""
    frames = list(frames)
    if mean is None:
        mean = np.zeros(frames[0].shape, dtype=np.float64)                       # :181
        m2 = np.zeros(frames[0].shape, dtype=np.float64) if use_std else None    # :184
    C = mean.shape[-1]
    for frame in frames:
        count += 1                                                               # :197
        if icrf is not None:
            f = icrf[frame, np.arange(C)]                                        # :201
        else:
            f = (frame / MAX_DN).astype(np.float64)                              # :203
        delta = f - mean                                                         # :205
        mean = mean + delta / count                                              # :206
        if use_std:
            m2 = m2 + delta * (f - mean)                                         # :208
    return mean, m2, count


def welford_finalize(mean, m2, count):
    """video_processing.py:210-215 as written (the std frame is not rescaled by MAX_DN)."""
    mean_u8 = np.around(mean * MAX_DN).astype(np.uint8)
    std_u8 = None
    if m2 is not None:
        std_u8 = np.around(np.sqrt(m2 / (count - 1)) / np.sqrt(count)).astype(np.uint8)
    return mean_u8, std_u8


def welford(frames, icrf=None, use_std=True):
    mean, m2, count = welford_state(frames, icrf, use_std)
    mean_u8, std_u8 = welford_finalize(mean, m2, count)
    return {"mean": mean_u8, "std": std_u8}


# --------------------------------------------------------------------------------------------
# ICRF-calibration energy function (modules/ICRF_calibration_exposure.py:22-201)
# --------------------------------------------------------------------------------------------
def candidate_icrf(mean_icrf, pca_array, params, use_mean_icrf=True, bits: int = BITS):
    """_inverse_camera_response_function (:22-45) followed by the shift of _energy_function (:166-167)."""
    params = np.asarray(params, dtype=np.float64)
    if use_mean_icrf:
        icrf = mean_icrf + np.matmul(pca_array, params)                          # :41-43
    else:
        icrf = np.linspace(0, 1, bits) ** params[0] + np.matmul(pca_array, params[1:])   # :38-39
    icrf = icrf + (1 - icrf[-1])                                                 # :166
    icrf[0] = 0                                                                  # :167
    return icrf


def candidate_valid(icrf) -> bool:
    """The rejection tests of _energy_function (:173-179): range, then strict monotonicity."""
    if np.max(icrf) > 1 or np.min(icrf) < 0:
        return False
    return bool(np.all(icrf[1:] > icrf[:-1]))


def analyze_linearity_pairs(values, stds, lower, upper, use_relative, exposures):
    """analyze_linearity (:66-145) pair by pair: values / stds are (X, Y, N) float64 (stds may be None); returns the
    N(N-1)/2 results in np.triu_indices(N, 1) order. Same per-element arithmetic as the (X, Y, N, N) broadcast."""
    X, Y, N = values.shape
    masked = np.where((values < lower) | (values > upper), np.nan, values)       # :96-97
    out = []
    with np.errstate(all="ignore"):
        for i in range(N):
            for j in range(i + 1, N):
                ratio = exposures[i] / exposures[j]                              # :100
                vi, vj = masked[:, :, i], masked[:, :, j]
                scaled = vj * ratio                                              # :111
                d = vi - scaled                                                  # :114
                if use_relative:
                    d = d / scaled                                               # :117
                a = np.abs(d)                                                    # :120
                if stds is not None:
                    si, sj = stds[:, :, i], stds[:, :, j]
                    if use_relative:
                        sigma = np.sqrt((si / scaled) ** 2 + ((vi * sj) / (ratio * vj ** 2)) ** 2)   # :127
                    else:
                        sigma = np.sqrt(si ** 2 + (ratio * sj) ** 2)             # :129
                    finite = np.logical_and(np.isfinite(a), sigma != 0)          # :133
                    w = np.where(finite, 1 / sigma, np.nan)                      # :134
                    ok = ~np.isnan(a) & ~np.isnan(w)                             # general_functions.py:164
                    num = np.nansum(a * w * ok)                                  # :167
                    den = np.nansum(ok * w)                                      # :170
                    out.append(num / den if den != 0 else np.nan)                # :173-174
                else:
                    out.append(np.nanmean(a) if np.any(~np.isnan(a)) else np.nan)   # :138
    return np.array(out)


def energy_function(icrf, dn_stack, std_stack, lower, upper, exposures):
    """_energy_function (:148-201) from the shifted candidate ICRF on: +inf for rejected candidates, the
    LUT-mapped stack through analyze_linearity (relative), nanmean over pairs, NaN -> +inf."""
    if not candidate_valid(icrf):
        return np.inf
    values = icrf[dn_stack]                                                      # :190
    res = analyze_linearity_pairs(values, std_stack, icrf[lower], icrf[upper], True, np.asarray(exposures, dtype=np.float64))
    with np.errstate(all="ignore"):
        e = np.nanmean(res) if np.any(~np.isnan(res)) else np.nan                # :196
    return float(np.inf if np.isnan(e) else e)                                   # :197-200


# --------------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md section 8(d) (shared by tests and the bench's cpu_baseline leg)
# --------------------------------------------------------------------------------------------
def synthetic_stack(seed: int, n: int, h: int, w: int, c: int = 3, with_std: bool = False):
    rng = np.random.default_rng(seed)
    rad = rng.random((h, w, c)) * 4
    t = 1e-3 * 2.0 ** np.arange(n)
    k = 255 / (4 * t[n // 2])
    frames = [np.clip(np.around(rad * ti * k), 0, 255).astype(np.uint8) for ti in t]
    stds = [0.004 * (1 + rng.random((h, w, c))) for _ in range(n)] if with_std else None
    return frames, stds, t


def synthetic_icrf(gammas=(2.2, 2.0, 1.8), bits: int = BITS):
    icrf = np.stack([np.linspace(0, 1, bits) ** g for g in gammas], axis=1)
    return icrf, icrf_derivative(icrf, bits)
