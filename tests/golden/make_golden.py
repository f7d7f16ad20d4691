#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Runs only in the build container, where the upstream reference is mounted
read-only at /root/reference. Nothing from the reference is copied: its modules
are imported in place (with two in-memory shims: a stub `cv2` - the reference
only uses it for file IO - and a stub `read_config` that supplies the values the
gitignored data/config.ini would hold), evaluated on seeded synthetic inputs,
and the inputs + outputs are written as plain float64/uint8 arrays (.npz,
loadable with allow_pickle=False).

What is evaluated (SURVEY.md section 8c - the reference's merge loop does not run
as written at HEAD, so the oracle is composed from the reference primitives that
do run, with the documented deviations A-J only):

  1. frame -> Measurand(); m.val = u8.astype(f64)/255 (modules/image_set.py:223)
  2. w, dw = m.apply_gaussian_weight()                (modules/measurand.py:606-618, as written)
  3. g_c, dg_c = m.extract(c, -1).linearize(ICRF[:, c], ICRF_diff[:, c])
                                                      (modules/measurand.py:352-373, 516-541, as written)
     cross-check (val only): einsum('hwcc->hwc') of the as-written _linearize_channel (:487-514)
  4. S = sum_i w_i through Measurand.__add__ (:106), S2 = S ** 2 through __pow__ (:217)
  5. modules/exposure_series.py:388-389 and :394 verbatim on arrays, ascending exposure
  6. hot pixel filter = where(dark > thr, scipy.ndimage.median_filter(x, (k,k), axes=(0,1), 'reflect'), x)
     (intended semantics of modules/measurand.py:543-557), flat field = the formulas of
     modules/measurand.py:582-602 with the integer ROI of deviation H.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import os
import pathlib
import sys
import types

import numpy as np

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent

SETTINGS = {
    "image size x": 32, "image size y": 48, "channels": 3, "bit depth": 8,
    "final datapoints": 256, "datapoint multiplier": 1, "original DoRF datapoints": 1024,
    "number of principal components": 5, "dark threshold": 0.05,
    "flat field middle zone percentage": 0.2, "hot pixel threshold": 0.02,
    "median filter kernel size": 3, "lower linearity limit": 5, "upper linearity limit": 250,
}


def import_reference():
    if not REF.is_dir():
        raise SystemExit("make_golden.py needs the reference mounted at /root/reference")
    sys.dont_write_bytecode = True
    cv2 = types.ModuleType("cv2")
    for name in ("imread", "imwrite", "VideoCapture", "imshow", "namedWindow", "waitKey", "destroyAllWindows"):
        setattr(cv2, name, lambda *a, **k: None)
    cv2.IMREAD_UNCHANGED = -1
    cv2.WINDOW_NORMAL = 0
    sys.modules["cv2"] = cv2
    rc = types.ModuleType("read_config")
    rc.current_directory = REF / "modules"
    rc.root_directory = REF
    rc.data_directory = REF / "data"
    rc.read_config_single = lambda key: SETTINGS.get(key, "unused")
    rc.read_config_list = lambda key: ["Blue", "Green", "Red"] if key == "channel names" else ["unused"]
    sys.modules["read_config"] = rc
    sys.path.insert(0, str(REF / "modules"))
    import measurand  # noqa
    import measurand_factory  # noqa
    import image_set  # noqa
    import exposure_series  # noqa
    return measurand, measurand_factory, image_set, exposure_series


ref_measurand, ref_factory, ref_image_set, ref_exposure_series = import_reference()
from scipy.ndimage import median_filter  # noqa: E402  (third-party call site measurand.py:21,546,553)

Measurand = lambda v=None, s=None: ref_factory.Measurand(v, s, use_cupy=False)  # noqa: E731


def make_icrf(gammas, bits=256):
    """ICRF fixture recipe of tests/unit/test_measurand.py:14-23 (dx = 2/(BITS-1))."""
    icrf = np.empty((bits, len(gammas)))
    diff = np.empty((bits, len(gammas)))
    for c, gma in enumerate(gammas):
        icrf[:, c] = np.linspace(0, 1, bits) ** gma
        diff[:, c] = np.gradient(icrf[:, c], 2 / (bits - 1))
    return icrf, diff


def synth_stack(rng, n, h, w, c=3):
    rad = rng.random((h, w, c)) * 4
    t = 1e-3 * 2.0 ** np.arange(n)
    k = 255 / (4 * t[n // 2])
    frames = np.stack([np.clip(np.around(rad * ti * k), 0, 255).astype(np.uint8) for ti in t])
    return frames, t


def ref_linearize(m, icrf, icrf_diff):
    """Per-channel _linearize_single through the reference (deviation A/B)."""
    C = m.val.shape[-1]
    vals, stds = [], []
    for c in range(C):
        mc = m.extract(c, axis=-1)
        lin = mc.linearize(icrf[:, c], None if icrf_diff is None else icrf_diff[:, c])
        vals.append(lin.val)
        stds.append(lin.std)
    val = np.concatenate(vals, axis=-1)
    std = None if stds[0] is None else np.concatenate(stds, axis=-1)
    return val, std


def ref_index(v):
    """The index expression of measurand.py:503/531 as written."""
    return np.around(v * 255).astype(np.dtype("uint8"))


def hot_filter(x, dark_val, thr, k):
    med = median_filter(x, size=(k, k), axes=(0, 1), mode="reflect")
    return np.where(dark_val > thr, med, x)


def roi_bounds(size_x, size_y, p):
    dx = int(np.floor(size_x * p))
    dy = int(np.floor(size_y * p))
    i = (int(np.floor(1 / p)) - 1) // 2
    return i * dx, (i + 1) * dx, i * dy, (i + 1) * dy


def flat_field(val, std, fval, fstd, size_x, size_y, p):
    """measurand.py:582-602 with the integer ROI (deviation H)."""
    x0, x1, y0, y1 = roi_bounds(size_x, size_y, p)
    m = np.mean(fval[x0:x1, y0:y1, :], axis=(0, 1))
    s = np.mean(fstd[x0:x1, y0:y1, :], axis=(0, 1))
    u_acq = (std ** 2) / (fval ** 2)
    u_acq *= m ** 2
    u_ff = (val ** 2) / (fval ** 4)
    u_ff *= fstd ** 2
    u_ff *= m ** 2
    u_ffm = (val ** 2) / (fval ** 2)
    u_ffm *= s ** 2
    ret_std = np.sqrt(u_acq + u_ff + u_ffm)
    ret_val = (val / fval) * m
    return ret_val, ret_std, m, s


def merge(frames_val, stds, t, icrf, icrf_diff, hot=None):
    """frames_val: list of f64 (H,W,C) arrays in [0,1] (or arbitrary floats); stds list or None.
    hot: None or list of per-frame (dark_val f64 | None), thr, k."""
    n = len(frames_val)
    ms = []
    for i in range(n):
        v = frames_val[i]
        s = None if stds is None else stds[i]
        if hot is not None and hot[0][i] is not None:
            v = hot_filter(v, hot[0][i], hot[1], hot[2])
            if s is not None:
                s = hot_filter(s, hot[0][i], hot[1], hot[2])
        m = Measurand()
        m.val = v
        if s is not None:
            m.std = s
        ms.append(m)
    # pass 1: sum of weights through the reference operators
    S = Measurand(np.zeros_like(frames_val[0]))
    for m in ms:
        S = S + m.apply_gaussian_weight()[0]
    S2 = (S ** 2).val
    S = S.val
    hdr_val = np.zeros_like(frames_val[0])
    hdr_std = None if stds is None else np.zeros_like(frames_val[0])
    idxs = []
    for i, m in enumerate(ms):
        w, dw = m.apply_gaussian_weight()
        idxs.append(ref_index(m.val))
        g, dg = ref_linearize(m, icrf, icrf_diff if stds is not None else None)
        hdr_val += (w * g) / (S * t[i])
        if stds is not None:
            hdr_std += (((dw * g + w * dg) / S - (dw * w * g) / S2) * dg / t[i]) ** 2
    if hdr_std is not None:
        hdr_std = hdr_std ** (1 / 2)
    return dict(val=hdr_val, std=hdr_std, S=S, idx=np.stack(idxs))


def save(name, **arrays):
    arrays = {k: v for k, v in arrays.items() if v is not None}
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"wrote {name}.npz  ({os.path.getsize(OUT / (name + '.npz')) / 1024:.0f} KiB)")


def case_identity():
    """(i) 3x16x16x3, identity ICRF, val only (config-1 shape in miniature)."""
    rng = np.random.default_rng(0)
    frames, t = synth_stack(rng, 3, 16, 16)
    icrf, diff = make_icrf((1.0, 1.0, 1.0))
    out = merge([f.astype(np.float64) / 255 for f in frames], None, t, icrf, diff)
    save("merge_identity", frames=frames, exposures=t, icrf=icrf, icrf_diff=diff,
         val=out["val"], S=out["S"], idx=out["idx"])


def case_std():
    """(ii) 3x32x32x3, ICRF linspace**(c+1), std = 0.1*v."""
    rng = np.random.default_rng(1)
    frames, t = synth_stack(rng, 3, 32, 32)
    icrf, diff = make_icrf((1.0, 2.0, 3.0))
    fv = [f.astype(np.float64) / 255 for f in frames]
    stds = [0.1 * v for v in fv]
    out = merge(fv, stds, t, icrf, diff)
    # as-written cross-check of the multi-channel gather (val only, deviation A)
    m = Measurand()
    m.val = fv[1]
    diag = np.einsum("hwcc->hwc", m.linearize(icrf).val)
    lin_val, lin_std = ref_linearize(Measurand(fv[1], stds[1]), icrf, diff)
    assert np.array_equal(diag, lin_val)
    save("merge_std", frames=frames, stds=np.stack(stds), exposures=t, icrf=icrf, icrf_diff=diff,
         val=out["val"], std=out["std"], S=out["S"], idx=out["idx"],
         lin1_val=lin_val, lin1_std=lin_std)


def case_full():
    """(iii) 7x32x48x3 radiance stack + std + flat + darks with hot pixels (config-3 shape in miniature)."""
    rng = np.random.default_rng(2)
    n, h, w = 7, 32, 48
    frames, t = synth_stack(rng, n, h, w)
    stds = [0.004 * (1 + rng.random((h, w, 3))) for _ in range(n)]
    icrf, diff = make_icrf((2.2, 2.0, 1.8))
    flat_u8 = np.clip(np.around(255 * (0.8 + 0.05 * rng.random((h, w, 3)))), 0, 255).astype(np.uint8)
    flat_std = 0.002 * (1 + 0.1 * rng.random((h, w, 3)))
    # dark frames at 16, 32 and 64 ms: dim background + injected hot pixels (also at edges/corners)
    def dark(density):
        d = rng.integers(0, 4, size=(h, w, 3)).astype(np.uint8)
        hotmask = rng.random((h, w, 3)) < density
        d[hotmask] = rng.integers(40, 256, size=int(hotmask.sum())).astype(np.uint8)
        d[0, 0, 0] = 200; d[h - 1, w - 1, 2] = 180; d[0, w // 2, 1] = 90; d[h // 2, 0, 1] = 250
        d[h - 1, 3, 0] = 77; d[5, w - 1, 2] = 99
        return d
    dark16, dark32, dark64 = dark(0.02), dark(0.025), dark(0.03)
    dark_exp = np.array([0.016, 0.032, 0.064])
    thr, k = 0.012, 3
    # dark selection of image_set.py:157-198 with the scale fix of deviation I:
    # t >= thr -> exact-exposure dark, else the next longer dark scaled by t/t_dark
    darks_val = []
    dark_sel = []
    for ti in t:
        if ti < thr:
            darks_val.append(None); dark_sel.append(-1)
            continue
        if np.any(dark_exp == ti):
            j = int(np.nonzero(dark_exp == ti)[0][0]); sc = 1.0
        else:
            raise AssertionError("case_full uses exact-exposure darks only")
        d = (dark16, dark32, dark64)[j].astype(np.float64) / 255 * sc
        darks_val.append(d); dark_sel.append(j)
    fv = [f.astype(np.float64) / 255 for f in frames]
    out = merge(fv, stds, t, icrf, diff, hot=(darks_val, thr, k))
    out_nohot = merge(fv, stds, t, icrf, diff)
    ffv, ffs, m, s = flat_field(out["val"], out["std"], flat_u8.astype(np.float64) / 255, flat_std, h, w, 0.2)
    save("merge_full", frames=frames, stds=np.stack(stds), exposures=t, icrf=icrf, icrf_diff=diff,
         dark16=dark16, dark32=dark32, dark64=dark64, dark_exposures=dark_exp, dark_sel=np.array(dark_sel),
         dark_threshold=np.array(thr), median_k=np.array(k),
         flat=flat_u8, flat_std=flat_std, ff_mid=np.array(0.2), ff_mean=m, ff_std_mean=s,
         val=out["val"], std=out["std"], S=out["S"], idx=out["idx"],
         val_nohot=out_nohot["val"], std_nohot=out_nohot["std"],
         val_ff=ffv, std_ff=ffs)


def case_dark_scaled():
    """(iii-b) dark selection with scaling: 3 frames at 20/40/80 ms, darks at 10/50/100 ms, thr 0.03."""
    rng = np.random.default_rng(5)
    h, w = 12, 20
    t = np.array([0.020, 0.040, 0.080])
    rad = rng.random((h, w, 3)) * 4
    kk = 255 / (4 * t[1])
    frames = np.stack([np.clip(np.around(rad * ti * kk), 0, 255).astype(np.uint8) for ti in t])
    stds = [0.004 * (1 + rng.random((h, w, 3))) for _ in range(3)]
    icrf, diff = make_icrf((2.2, 2.0, 1.8))
    dark_exp = np.array([0.010, 0.050, 0.100])
    darks = []
    for _ in dark_exp:
        d = rng.integers(0, 6, size=(h, w, 3)).astype(np.uint8)
        hm = rng.random((h, w, 3)) < 0.05
        d[hm] = rng.integers(8, 256, size=int(hm.sum())).astype(np.uint8)
        darks.append(d)
    thr, k = 0.03, 3
    # image_set.py:173-196: walk the dark list in order; return the first dark with exposure > t once a shorter one
    # has been seen too ("greater_index" = last greater seen so far), scaled by t/t_dark (deviation I).
    darks_val, sel, scales = [], [], []
    for ti in t:
        lesser = greater = False
        gi = 0
        chosen = None
        if ti >= thr:
            for i, de in enumerate(dark_exp):
                if de < ti:
                    lesser = True
                if de > ti:
                    greater = True; gi = i
                if de == ti:
                    chosen = (i, 1.0); break
                if lesser and greater:
                    chosen = (gi, ti / dark_exp[gi]); break
        if chosen is None:
            darks_val.append(None); sel.append(-1); scales.append(0.0)
        else:
            # scale through the reference operator: (scale) * measurand == Measurand.__rmul__
            dm = Measurand(darks[chosen[0]].astype(np.float64) / 255)
            scaled = chosen[1] * dm if chosen[1] != 1.0 else dm
            darks_val.append(scaled.val); sel.append(chosen[0]); scales.append(chosen[1])
    fv = [f.astype(np.float64) / 255 for f in frames]
    out = merge(fv, stds, t, icrf, diff, hot=(darks_val, thr, k))
    save("merge_dark_scaled", frames=frames, stds=np.stack(stds), exposures=t, icrf=icrf, icrf_diff=diff,
         darks=np.stack(darks), dark_exposures=dark_exp, dark_sel=np.array(sel), dark_scale=np.array(scales),
         dark_threshold=np.array(thr), median_k=np.array(k),
         val=out["val"], std=out["std"], S=out["S"], idx=out["idx"])


def case_ramp():
    """(iv) every DN 0..255 in every channel, incl. saturated 0 and 255: idx / LUT exactness."""
    dn = np.arange(256, dtype=np.uint8)
    base = np.stack([dn, dn[::-1], np.roll(dn, 77)], axis=-1)            # (256,3)
    frames = np.stack([np.tile(base[None], (4, 1, 1)),
                       np.tile(np.roll(base, 13, axis=0)[None], (4, 1, 1)),
                       np.tile(np.full_like(base, 255)[None], (4, 1, 1)),
                       np.tile(np.zeros_like(base)[None], (4, 1, 1))])     # (4, 4, 256, 3)
    t = np.array([0.001, 0.004, 0.016, 0.064])
    icrf, diff = make_icrf((2.2, 2.0, 1.8))
    fv = [f.astype(np.float64) / 255 for f in frames]
    rng = np.random.default_rng(3)
    stds = [0.004 * (1 + rng.random(f.shape)) for f in fv]
    out = merge(fv, stds, t, icrf, diff)
    m = Measurand(); m.val = fv[0]
    w, dw = m.apply_gaussian_weight()
    save("merge_ramp", frames=frames, stds=np.stack(stds), exposures=t, icrf=icrf, icrf_diff=diff,
         val=out["val"], std=out["std"], S=out["S"], idx=out["idx"],
         w_lut=w[0, :, 0], dw_lut=dw[0, :, 0])


def case_float():
    """(v) float64 input: .5 ties of v*255 (half-to-even), values off the k/255 grid, >1.0 wrap, and NEGATIVE values in [-3/255, 0)
    (bias-subtracted frames, modules/image_set.py:520,538): around() then astype(uint8) wraps them to 253..255 (modules/measurand.py:503,531)."""
    rng = np.random.default_rng(4)
    h, w = 8, 24
    v = rng.random((3, h, w, 3))
    ties = (np.arange(0, 255)[:72] + 0.5) / 255                           # exact .5 products where representable
    v[0].reshape(-1)[:72] = ties
    v[1].reshape(-1)[:32] = 1.0 + rng.random(32) * 0.004                  # rounds to 255 or 256 (wraps to 0)
    v[2].reshape(-1)[:16] = np.arange(16) / 255                           # exact grid values
    v[2].reshape(-1)[16:24] = 1.0
    v[2].reshape(-1)[24:32] = 0.0
    # negatives (no further rng draws, so every other entry of the fixture keeps its value): -0.25/255 .. -3/255 in steps of 0.25/255 - this
    # holds the ties -0.5/255 (-> -0 -> 0), -1.5/255 and -2.5/255 (-> -2 -> 254) - plus -1e-9, -0.0 and the exact -1/255, -2/255, -3/255
    v[1].reshape(-1)[40:52] = -np.arange(1, 13) * 0.25 / 255
    v[1].reshape(-1)[52:57] = [-1e-9, -0.0, -1 / 255, -2 / 255, -3 / 255]
    v[0].reshape(-1)[80:84] = [-0.5 / 255, -1.5 / 255, -2.5 / 255, -2.9 / 255]
    t = np.array([0.002, 0.004, 0.008])
    icrf, diff = make_icrf((2.2, 2.0, 1.8))
    stds = [0.01 * (1 + rng.random((h, w, 3))) for _ in range(3)]
    out = merge([v[i] for i in range(3)], stds, t, icrf, diff)
    assert v.min() < -2.9 / 255 and set(np.unique(out["idx"][1].reshape(-1)[40:57])) >= {0, 253, 254, 255}
    m = Measurand(v[0].copy(), stds[0].copy())
    wgt, dwgt = m.apply_gaussian_weight()
    lin_val, lin_std = ref_linearize(m, icrf, diff)
    save("merge_float", frames_f64=v, stds=np.stack(stds), exposures=t, icrf=icrf, icrf_diff=diff,
         val=out["val"], std=out["std"], S=out["S"], idx=out["idx"],
         w0=wgt, dw0=dwgt, lin0_val=lin_val, lin0_std=lin_std)


def case_operators():
    """Row 10: Measurand operators with first-order propagation, from the reference classes as written."""
    rng = np.random.default_rng(6)
    a = rng.random((4, 5, 3)) + 0.25
    b = rng.random((5, 3)) + 0.25
    sa, sb = 0.1 * a, 0.05 * b
    out = dict(a=a, b=b, sa=sa, sb=sb)
    combos = {"ss": (sa, sb), "sn": (sa, None), "ns": (None, sb), "nn": (None, None)}
    for tag, (s1, s2) in combos.items():
        A, B = Measurand(a, s1), Measurand(b, s2)
        for opname, fn in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y),
                           ("div", lambda x, y: x / y), ("pow", lambda x, y: x ** y)):
            r = fn(A, B)
            out[f"{opname}_{tag}_val"] = r.val
            if r.std is not None:
                out[f"{opname}_{tag}_std"] = r.std
    A = Measurand(a, sa)
    for opname, r in (("neg", -A), ("loge", A.log_e()), ("log10", A.log_10()), ("rmul", 2.5 * A),
                      ("adds", A + 1.5), ("subs", A - 0.125), ("muls", A * 3.0), ("divs", A / 4.0), ("pows", A ** 2),
                      ("sqrt", A ** (1 / 2))):
        out[f"{opname}_val"] = r.val
        out[f"{opname}_std"] = r.std
    An = Measurand(a)
    for opname, r in (("pows_n", An ** 2), ("muls_n", An * 3.0)):
        out[f"{opname}_val"] = r.val
    # gaussian weight + zeros_like + extract
    w, dw = A.apply_gaussian_weight()
    out["gw_w"], out["gw_dw"] = w, dw
    ex = A.extract([0, 2], axis=-1)
    out["extract_val"], out["extract_std"] = ex.val, ex.std
    # compute_difference / interpolate (callers: image_set.py:438-480)
    Bfull = Measurand(rng.random((4, 5, 3)) + 0.25, 0.07 * np.ones((4, 5, 3)))
    out["b2"], out["sb2"] = Bfull.val, Bfull.std
    ad, rd = ref_measurand.AbstractMeasurand.compute_difference(A, Bfull, 0.5)
    out["cd_abs_val"], out["cd_abs_std"], out["cd_rel_val"], out["cd_rel_std"] = ad.val, ad.std, rd.val, rd.std
    ip = ref_measurand.AbstractMeasurand.interpolate(A, Bfull, 1.0, 3.0, 1.5)
    out["interp_val"], out["interp_std"] = ip.val, ip.std
    # statistics with and without std, nan handling
    a_nan = a.copy(); a_nan[0, 1, 2] = np.nan; a_nan[3, 4, 0] = np.nan
    sa_nan = sa.copy(); sa_nan[0, 1, 2] = np.nan; sa_nan[3, 4, 0] = np.nan
    st = Measurand(a_nan.copy(), sa_nan.copy()).compute_dimension_statistics(axis=(0, 1))
    out["stat_s_mean"], out["stat_s_std"], out["stat_s_err"] = st["mean"], st["std"], st["error"]
    st = Measurand(a_nan.copy()).compute_dimension_statistics(axis=(0, 1))
    out["stat_n_mean"], out["stat_n_std"] = st["mean"], st["std"]
    out["a_nan"], out["sa_nan"] = a_nan, sa_nan
    # apply_thresholds (in place)
    T = Measurand(a.copy(), sa.copy())
    T.apply_thresholds([0.4, None, 0.5], [1.0, 0.9, None])
    out["thr_val"], out["thr_std"] = T.val, T.std
    save("operators", **out)


def case_welford():
    """welford_algorithm (modules/video_processing.py:161-219) run as written on a seeded uint8 clip,
    ICRF=None path (the ICRF path raises at HEAD: `if ICRF:` on an ndarray, deviation K)."""
    import cv2
    import general_functions as gf
    import video_processing as vp
    rng = np.random.default_rng(11)
    h, w, n = 12, 20, 37
    base = rng.integers(0, 256, (h, w, 3))
    clip = np.clip(base[None] + np.around(rng.standard_normal((n, h, w, 3)) * 6), 0, 255).astype(np.uint8)

    class Capture:                                     # stands in for cv.VideoCapture: only the two size queries
        def __init__(self, path): pass
        def get(self, prop): return {3: w, 4: h}[prop]
    cv2.VideoCapture = Capture
    cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT = 3, 4

    def frames_of(path):
        for f in clip:
            yield f
        yield None
    gf.video_frame_generator = frames_of
    with np.errstate(all="ignore"):
        r_std = vp.welford_algorithm(pathlib.Path("clip.avi"), None, True)
        r_mean = vp.welford_algorithm(pathlib.Path("clip.avi"), None, False)
    assert r_mean["std"] is None and np.array_equal(r_mean["mean"], r_std["mean"])
    save("welford", clip=clip, mean=r_std["mean"], std=r_std["std"])


def case_energy():
    """_energy_function / analyze_linearity (modules/ICRF_calibration_exposure.py:66-201) as written,
    on a seeded (X, Y, N) uint8 channel stack, with and without std, for several PCA coefficient vectors
    (including candidates the range / monotonicity tests reject)."""
    import ICRF_calibration_exposure as ice
    rng = np.random.default_rng(12)
    X, Y, N = 14, 18, 5
    t = 1e-3 * 2.0 ** np.arange(N)
    rad = rng.random((X, Y)) * 4
    k = 255 / (4 * t[N // 2])
    true_resp = np.linspace(0, 1, 256) ** 1.8
    lin = np.clip(rad[..., None] * t * k / 255, 0, 1)
    dn = np.clip(np.around(np.interp(lin, true_resp, np.linspace(0, 1, 256)) * 255 + rng.standard_normal((X, Y, N))), 0, 255).astype(np.uint8)
    sd = 0.004 * (1 + rng.random((X, Y, N)))
    sd[0, 0, :] = 0.0                                   # zero std: sigma == 0 -> excluded (:133)
    mean_icrf = np.linspace(0, 1, 256) ** 2.0
    xs = np.linspace(0, 1, 256)
    pca = np.stack([np.sin(np.pi * (m + 1) * xs) * 0.05 / (m + 1) for m in range(5)], axis=1)
    params = np.concatenate([np.zeros((1, 5)), rng.uniform(-1, 1, (9, 5)), rng.uniform(-8, 8, (4, 5))])
    lower, upper = 5, 250
    e_plain, e_std, pairs_plain, pairs_std, icrfs = [], [], [], [], []
    with np.errstate(all="ignore"):
        for pv in params:
            e_plain.append(ice._energy_function(pv.copy(), mean_icrf.copy(), pca, dn, None, lower, upper, True, t))
            e_std.append(ice._energy_function(pv.copy(), mean_icrf.copy(), pca, dn, sd, lower, upper, True, t))
            c = ice._inverse_camera_response_function(mean_icrf.copy(), pca, pv.copy(), True)
            c += 1 - c[-1]
            c[0] = 0
            icrfs.append(c)
            vals = c[dn]
            pairs_plain.append(ice.analyze_linearity(vals, None, c[lower], c[upper], True, t))
            pairs_std.append(ice.analyze_linearity(vals, sd, c[lower], c[upper], True, t))
        vals = icrfs[0][dn]
        abs_plain = ice.analyze_linearity(vals, None, icrfs[0][lower], icrfs[0][upper], False, t)
        abs_std = ice.analyze_linearity(vals, sd, icrfs[0][lower], icrfs[0][upper], False, t)
    save("energy", dn=dn, sd=sd, exposures=t, mean_icrf=mean_icrf, pca=pca, params=params,
         lower=np.array(lower), upper=np.array(upper), icrfs=np.stack(icrfs),
         energy_plain=np.array(e_plain), energy_std=np.array(e_std),
         pairs_plain=np.stack(pairs_plain), pairs_std=np.stack(pairs_std), abs_plain=abs_plain, abs_std=abs_std)


def case_helpers():
    """The host-side helpers of modules/general_functions.py that the path's callers use (exposure_series.py:434 map_linearity_limits,
    ICRF_calibration_exposure.py choose_evenly_spaced_points / predict_output_shape / nanaverage / weighted_avg_and_std, measurand.py
    is_broadcastable, weighted_percentile), run as written on seeded inputs."""
    import general_functions as gf
    rng = np.random.default_rng(21)
    out = {}
    shapes = [((3, 4, 5), (4, 5)), ((3, 4, 5), (3, 1, 5)), ((2, 3), (3, 2)), ((7,), (1,)), ((5, 1, 6), (4, 6)), ((5, 2, 6), (4, 6))]
    out["bc_shapes_a"] = np.array([list(a) + [0] * (3 - len(a)) for a, _ in shapes]); out["bc_len_a"] = np.array([len(a) for a, _ in shapes])
    out["bc_shapes_b"] = np.array([list(b) + [0] * (3 - len(b)) for _, b in shapes]); out["bc_len_b"] = np.array([len(b) for _, b in shapes])
    out["bc_result"] = np.array([gf.is_broadcastable(a, b) for a, b in shapes])
    img = rng.random((23, 31, 3))
    out["ces_in"] = img
    out["ces_5"] = gf.choose_evenly_spaced_points(img, 5)
    out["ces_4_7"] = gf.choose_evenly_spaced_points(img, 4, 7)
    out["pos"] = np.array([gf.predict_output_shape((23, 31), 5), gf.predict_output_shape((23, 31), 4, 7), gf.predict_output_shape((1, 1), 9)])
    v = rng.normal(size=200) * 2 + 1
    w = rng.random(200) + 0.1
    out["was_v"], out["was_w"] = v, w
    out["was"] = np.array(gf.weighted_avg_and_std(v, w))
    out["was_none"] = np.array(gf.weighted_avg_and_std(v, None))
    a = rng.random((6, 7, 4)); a[rng.random(a.shape) < 0.2] = np.nan
    ww = rng.random((6, 7, 4)); ww[rng.random(ww.shape) < 0.2] = np.nan
    ww[:, 3, 1] = np.nan                                                     # a line without any valid weight -> NaN
    out["na_v"], out["na_w"] = a, ww
    with np.errstate(all="ignore"):
        out["na_axis0"] = gf.nanaverage(a, ww, 0)
        out["na_axis01"] = gf.nanaverage(a, ww, (0, 1))
        out["na_axis2"] = gf.nanaverage(a, ww, 2)
    pv = rng.normal(size=101)
    pw = rng.integers(1, 5, size=101).astype(np.float64)
    out["wp_v"], out["wp_w"] = pv, pw
    out["wp_default"] = gf.weighted_percentile(pv)
    out["wp_weighted"] = gf.weighted_percentile(pv, np.array([5.0, 50.0, 95.0]), pw)
    icrf = make_icrf((2.2, 2.0, 1.8))[0]
    out["mll_icrf"] = icrf
    lo, up = gf.map_linearity_limits(None, None, icrf); out["mll_none_icrf"] = np.stack([lo, up])
    lo, up = gf.map_linearity_limits(10, 20, icrf); out["mll_10_20_icrf"] = np.stack([lo, up])
    lo, up = gf.map_linearity_limits(None, None, None); out["mll_none_none"] = np.stack([lo, up])
    lo, up = gf.map_linearity_limits(7, 3, None); out["mll_7_3_none"] = np.stack([lo, up])
    save("helpers", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                                                    # python make_golden.py helpers [energy ...]: only those cases
        for name in sys.argv[1:]:
            globals()["case_" + name]()
        raise SystemExit(0)
    case_identity()
    case_std()
    case_full()
    case_dark_scaled()
    case_ramp()
    case_float()
    case_operators()
    case_welford()
    case_energy()
    case_helpers()
