"""Seeded random cases of the host build (libhdrmerge_host.so, Measurand(use_cupy=False)) against the NumPy oracle - the third side of the
triangle: the oracle is pinned to the reference's own outputs (tests/golden), the HIP library to the host build on hundreds of thousands of
random cases (tools/fuzz_backends.py, and bit for bit at BASELINE's full sizes), and here the host build to the oracle on random frame counts,
channel counts, ragged sizes, corrections and special values. CPU only; a few seconds."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import hdr_oracle as orc  # noqa: E402


def H(val=None, std=None):
    from camera_linearity_amd.measurand_factory import Measurand
    return Measurand(val, std, use_cupy=False)


@pytest.fixture(scope="module")
def heng():
    from camera_linearity_amd.measurand import _HOST_ENGINE
    return _HOST_ENGINE


def T(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x).copy())


def agree(got, ref, rtol, what, atol=0.0):
    got = got.numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bad_g, bad_r = ~np.isfinite(got), ~np.isfinite(ref)
    assert np.array_equal(bad_g, bad_r), (what, "non-finite positions")
    assert np.array_equal(got[bad_g], ref[bad_r], equal_nan=True), (what, "non-finite values")
    np.testing.assert_allclose(got[~bad_g], ref[~bad_r], rtol=rtol, atol=atol, err_msg=what)


def icrf_tables(rng, c):
    g = np.linspace(0, 1, 256)[:, None] ** rng.uniform(0.6, 2.6, size=c)[None, :]
    if rng.random() < 0.25:
        g = g - 0.1
    d = np.stack([np.gradient(g[:, k], 2 / 255) for k in range(c)], axis=1)
    return np.ascontiguousarray(g), np.ascontiguousarray(d)


@pytest.mark.parametrize("seed", range(40))
def test_random_merge(heng, seed):
    """hm_merge of the host build (one pass over any number of frames, nth_element medians, flat field last) against the oracle's
    frame-by-frame NumPy form: uint8 and float64 frames, 1-40 frames, 1-4 channels, std, per-frame dark maps with k = 3 / 5, flat field."""
    rng = np.random.default_rng(1000 + seed)
    c = int(rng.choice([1, 2, 3, 3, 4]))
    n = int(rng.choice([1, 2, 3, 5, 7, 9, 16, 17, 33, 40]))
    h, w = int(rng.integers(1, 24)), int(rng.integers(1, 40))
    f64 = rng.random() < 0.3
    with_std = rng.random() < 0.6
    t = np.sort(rng.uniform(1e-4, 2.0, size=n))
    frames = [rng.random((h, w, c)) * rng.choice([1.0, 1.15]) - rng.choice([0.0, 0.05]) for _ in range(n)] if f64 else \
             [rng.integers(0, 256, (h, w, c)).astype(np.uint8) for _ in range(n)]
    stds = [0.004 * (1 + rng.random((h, w, c))) for _ in range(n)] if with_std else None
    g, d = icrf_tables(rng, c)
    kw_h, kw_o = {}, {}
    if rng.random() < 0.5 and min(h, w) >= 1:
        k = int(rng.choice([3, 5]))
        darks = []
        for i in range(n):
            if rng.random() < 0.3:
                darks.append(None)
            else:
                dm = rng.integers(0, 20, (h, w, c)).astype(np.uint8)
                dm[rng.random(dm.shape) < rng.choice([0.01, 0.1])] = 220
                darks.append(dm)
        if any(x is not None for x in darks):
            kw_h = dict(darks=[T(x) for x in darks], dark_min=[100] * n, median_k=k)
            kw_o = dict(darks=[None if x is None else orc.unit_from_u8(x) for x in darks], dark_threshold=99.5 / 255, median_k=k)
    use_flat = rng.random() < 0.5
    if use_flat:
        flat = rng.integers(120, 250, (h, w, c)).astype(np.uint8)
        m, sm = rng.uniform(0.6, 0.9, size=c), rng.uniform(0.001, 0.003, size=c)
        fstd = 0.002 * (1 + rng.random((h, w, c)))
        kw_h.update(flat=T(flat), ff_mean=list(m))
        kw_o.update(flat=orc.unit_from_u8(flat), ff_mean=m)
        if with_std:
            kw_h.update(flat_std=T(fstd), ff_std_mean=list(sm))
            kw_o.update(flat_std=fstd, ff_std_mean=sm)
    got = heng.merge([T(f) for f in frames], list(t), g, d if with_std else None, None if stds is None else [T(s) for s in stds],
                     want_sum_w=True, **kw_h)
    if use_flat and not with_std:           # the oracle's normalize_by_map wants std operands: the value formula alone (measurand.py:602)
        kw_o.pop("flat"); kw_o.pop("ff_mean")
    with np.errstate(all="ignore"):
        ref = orc.merge(frames, t, g, d if with_std else None, stds, **kw_o)
    agree(got["sum_w"], ref["S"], 1e-13, "S")
    if use_flat and with_std:
        agree(got["val"], ref["val_ff"], 1e-11, "val_ff")
        agree(got["std"], ref["std_ff"], 1e-8, "std_ff", atol=1e-15)
    elif use_flat:
        agree(got["val"], (ref["val"] / orc.unit_from_u8(flat)) * m, 1e-11, "val / F * m")
    else:
        agree(got["val"], ref["val"], 1e-11, "val")
        if with_std:
            agree(got["std"], ref["std"], 1e-8, "std", atol=1e-15)


@pytest.mark.parametrize("seed", range(30))
def test_random_operators_and_unary(seed):
    rng = np.random.default_rng(2000 + seed)
    nd = int(rng.integers(1, 4))
    full = [int(rng.integers(1, 7)) for _ in range(nd)]
    a_shape = tuple(1 if rng.random() < 0.25 else s for s in full)
    b_shape = tuple(1 if rng.random() < 0.25 else s for s in full)
    x = rng.normal(size=a_shape) * 3
    y = rng.normal(size=b_shape) * 3
    sx = np.abs(rng.normal(size=a_shape)) * 0.05 if rng.random() < 0.6 else None
    sy = np.abs(rng.normal(size=b_shape)) * 0.05 if rng.random() < 0.6 else None
    if rng.random() < 0.4:
        for arr in (x, y):
            msk = rng.random(arr.shape) < 0.1
            arr[msk] = rng.choice([np.nan, np.inf, -np.inf, 0.0], size=int(msk.sum()))
    A, B = H(x.copy(), None if sx is None else sx.copy()), H(y.copy(), None if sy is None else sy.copy())
    with np.errstate(all="ignore"):
        for name, fn, ofn in (("add", lambda p, q: p + q, orc.op_add), ("sub", lambda p, q: p - q, orc.op_sub),
                              ("mul", lambda p, q: p * q, orc.op_mul), ("div", lambda p, q: p / q, orc.op_div)):
            r = fn(A, B)
            rv, rs = ofn(x, sx, y, sy)
            agree(r.val, rv, 1e-14, name)
            if rs is None:
                assert r.std is None
            else:
                agree(r.std, rs, 1e-13, name + ".std")
        xp = np.abs(x) + 0.1
        P = H(xp.copy(), None if sx is None else sx.copy())
        r = P ** B
        rv, rs = orc.op_pow(xp, sx, y, sy)
        agree(r.val, rv, 1e-12, "pow")
        if rs is not None:
            agree(r.std, rs, 1e-11, "pow.std")
        for name, fn, ofn in (("neg", lambda m_: -m_, orc.op_neg), ("log_e", lambda m_: m_.log_e(), orc.op_log_e),
                              ("log_10", lambda m_: m_.log_10(), orc.op_log_10)):
            r = fn(P)
            rv, rs = ofn(xp, sx)
            agree(r.val, rv, 1e-14, name)
            if rs is not None:
                agree(r.std, rs, 1e-13, name + ".std")


@pytest.mark.parametrize("seed", range(30))
def test_random_statistics_thresholds_differences(seed):
    rng = np.random.default_rng(3000 + seed)
    c = int(rng.choice([1, 3, 4]))
    shape = (int(rng.integers(1, 20)), int(rng.integers(1, 20)), c)
    x = rng.random(shape) * 4 + 0.05
    y = rng.random(shape) * 4 + 0.05
    x[rng.random(shape) < 0.1] = np.nan
    if rng.random() < 0.4:
        x[rng.random(shape) < 0.03] = np.inf
        y[rng.random(shape) < 0.03] = 0.0
    with_std = rng.random() < 0.5
    sx = 0.01 + 0.05 * rng.random(shape) if with_std else None
    sy = 0.01 + 0.05 * rng.random(shape) if with_std else None
    with np.errstate(all="ignore"):
        for axis in ((0, 1), 0, None) + (() if with_std else (1, 2, (0, 2), (1, 2))):
            got = H(x.copy(), None if sx is None else sx.copy()).compute_dimension_statistics(axis)
            ref = orc.dimension_statistics(x, sx, axis)
            mag = float(np.nanmax(np.abs(x[np.isfinite(x)]))) if np.isfinite(x).any() else 1.0
            agree(got["mean"], ref["mean"], 1e-11, f"mean axis={axis}")
            agree(got["std"], ref["std"], 1e-9, f"std axis={axis}", atol=1e-13 * mag)
            if with_std:
                agree(got["error"], ref["error"], 1e-12, f"error axis={axis}")
        lo = [float(rng.uniform(0.0, 0.5)) if rng.random() < 0.7 else None for _ in range(c)]
        hi = [float(rng.uniform(2.0, 4.0)) if rng.random() < 0.7 else None for _ in range(c)]
        A = H(x.copy(), None if sx is None else sx.copy())
        A.apply_thresholds(lo, hi)
        tv, ts = orc.apply_thresholds(x.copy(), None if sx is None else sx.copy(), lo, hi)
        agree(A.val, tv, 0, "thresholds")
        if ts is not None:
            agree(A.std, ts, 0, "thresholds.std")
        mult = float(rng.uniform(0.2, 5))
        B = H(y.copy(), None if sy is None else sy.copy())
        ad, rd = type(A).compute_difference(A, B, mult)
        oa, oas, orl, ors = orc.compute_difference(tv, ts, y, sy, mult)
        agree(ad.val, oa, 1e-14, "abs diff")
        agree(rd.val, orl, 1e-13, "rel diff")
        if with_std:
            agree(ad.std, oas, 1e-13, "abs diff std")
            agree(rd.std, ors, 1e-12, "rel diff std")
        it = type(A).interpolate(A, B, 1.0, 3.0, 1.7)
        ov, os_ = orc.interpolate(tv, ts, y, sy, 1.0, 3.0, 1.7)
        agree(it.val, ov, 1e-13, "interpolate")
        if with_std:
            agree(it.std, os_, 1e-12, "interpolate std")


@pytest.mark.parametrize("seed", range(20))
def test_random_linearize_weights_corrections(heng, seed):
    rng = np.random.default_rng(4000 + seed)
    c = int(rng.choice([1, 3, 4]))
    h, w = int(rng.integers(5, 30)), int(rng.integers(5, 30))
    g, d = icrf_tables(rng, c)
    u8 = rng.random() < 0.5
    x = rng.integers(0, 256, (h, w, c)).astype(np.uint8) if u8 else rng.uniform(-0.3, 1.4, size=(h, w, c))
    if not u8:
        x = (np.round(x * 255) + rng.choice([0.0, 0.5], size=x.shape)) / 255
    s = np.abs(rng.normal(size=(h, w, c))) * 0.01
    val, std, idx = heng.linearize(T(x), T(s), g, d, return_index=True)
    xv = orc.unit_from_u8(x) if u8 else x
    rv, rs, ridx = orc.linearize(xv, s, g, d)
    assert np.array_equal(idx.numpy(), ridx)
    agree(val, rv, 0, "linearize")
    agree(std, rs, 1e-15, "linearize.std")
    wv, dwv = heng.gaussian_weight(T(x))
    rw, rdw = orc.gaussian_weight(xv)
    agree(wv, rw, 1e-14, "w")                       # (np.e ** x in the reference, exp(x) here: a few ulp on float64 values, none on the DN grid)
    agree(dwv, rdw, 1e-13, "dw", atol=1e-300)
    k = int(rng.choice([3, 5, 7]))
    dm = rng.integers(0, 30, (h, w, c)).astype(np.uint8)
    dm[rng.random(dm.shape) < 0.05] = 230
    f = heng.hot_pixel_filter(T(xv), T(orc.unit_from_u8(dm)), 0.5, k)
    agree(f, orc.hot_pixel_filter(xv, orc.unit_from_u8(dm), 0.5, k), 0, "hot pixel filter")
    flat = rng.integers(100, 250, (h, w, c)).astype(np.uint8)
    fstd = 0.002 * (1 + rng.random((h, w, c)))
    hdr, hs = rng.random((h, w, c)) * 4, np.abs(rng.normal(size=(h, w, c))) * 0.01
    m, sm = rng.uniform(0.6, 0.9, size=c), rng.uniform(0.001, 0.003, size=c)
    nv, ns = heng.normalize_by_map(T(hdr), T(hs), T(flat), T(fstd), m, sm)
    ov, os_ = orc.normalize_by_map(hdr, hs, orc.unit_from_u8(flat), fstd, m, sm)
    agree(nv, ov, 1e-14, "normalize")
    agree(ns, os_, 1e-13, "normalize.std")
