"""The oracle (oracle/hdr_oracle.py) against the vectors produced by running the reference
(tests/golden/make_golden.py). CPU only. Tolerance: the oracle keeps the reference's operation
order, so agreement is to a few ulps; 1e-13 relative leaves room for libm/SIMD `pow` differences
between the machine that generated the vectors and the one running the test."""
import numpy as np
import pytest

from oracle import hdr_oracle as orc

RTOL = 1e-13


def close(a, b, rtol=RTOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=0)


def test_weight_lut_matches_reference(golden):
    g = golden("merge_ramp")
    w, dw = orc.gaussian_weight_lut()
    close(w, g["w_lut"])
    close(dw, g["dw_lut"])


def test_index_identity_for_u8_frames():
    dn = np.arange(256, dtype=np.uint8)
    assert np.array_equal(orc.lut_index(orc.unit_from_u8(dn)), dn)


@pytest.mark.parametrize("name", ["merge_identity", "merge_std", "merge_ramp"])
def test_merge_u8(golden, name):
    g = golden(name)
    stds = list(g["stds"]) if "stds" in g else None
    out = orc.merge(list(g["frames"]), g["exposures"], g["icrf"], g["icrf_diff"], stds=stds)
    assert np.array_equal(out["idx"], g["idx"])
    close(out["S"], g["S"])
    close(out["val"], g["val"])
    if stds is not None:
        close(out["std"], g["std"])


def test_linearize_single_frame(golden):
    g = golden("merge_std")
    v = orc.unit_from_u8(g["frames"][1])
    val, std, idx = orc.linearize(v, g["stds"][1], g["icrf"], g["icrf_diff"])
    assert np.array_equal(idx, g["frames"][1])
    close(val, g["lin1_val"])
    close(std, g["lin1_std"])
    # integer input is used as the index directly (measurand.py:505)
    val2, _, idx2 = orc.linearize(g["frames"][1], None, g["icrf"])
    assert np.array_equal(idx2, g["frames"][1]) and np.array_equal(val2, val)


def test_merge_float_input(golden):
    g = golden("merge_float")
    out = orc.merge(list(g["frames_f64"]), g["exposures"], g["icrf"], g["icrf_diff"], stds=list(g["stds"]))
    assert np.array_equal(out["idx"], g["idx"])
    close(out["val"], g["val"])
    close(out["std"], g["std"])
    w, dw = orc.gaussian_weight(g["frames_f64"][0])
    close(w, g["w0"])
    close(dw, g["dw0"])
    val, std, _ = orc.linearize(g["frames_f64"][0], g["stds"][0], g["icrf"], g["icrf_diff"])
    close(val, g["lin0_val"])
    close(std, g["lin0_std"])


def _darks_for(g, dark_arrays):
    darks = []
    for i, t in enumerate(g["exposures"]):
        j, sc = orc.select_dark(float(t), [float(x) for x in g["dark_exposures"]], float(g["dark_threshold"]))
        assert j == int(g["dark_sel"][i])
        darks.append(None if j < 0 else orc.unit_from_u8(dark_arrays[j]) * sc if sc != 1.0
                     else orc.unit_from_u8(dark_arrays[j]))
    return darks


def test_merge_full_with_corrections(golden):
    g = golden("merge_full")
    darks = _darks_for(g, [g["dark16"], g["dark32"], g["dark64"]])
    h, w = g["flat"].shape[:2]
    fval = orc.unit_from_u8(g["flat"])
    m = orc.flat_roi_mean(fval, h, w, float(g["ff_mid"]))
    s = orc.flat_roi_mean(g["flat_std"], h, w, float(g["ff_mid"]))
    close(m, g["ff_mean"])
    close(s, g["ff_std_mean"])
    out = orc.merge(list(g["frames"]), g["exposures"], g["icrf"], g["icrf_diff"], stds=list(g["stds"]),
                    darks=darks, dark_threshold=float(g["dark_threshold"]), median_k=int(g["median_k"]),
                    flat=fval, flat_std=g["flat_std"], ff_mean=m, ff_std_mean=s)
    assert np.array_equal(out["idx"], g["idx"])
    close(out["val"], g["val"])
    close(out["std"], g["std"])
    close(out["val_ff"], g["val_ff"])
    close(out["std_ff"], g["std_ff"])
    out2 = orc.merge(list(g["frames"]), g["exposures"], g["icrf"], g["icrf_diff"], stds=list(g["stds"]))
    close(out2["val"], g["val_nohot"])
    close(out2["std"], g["std_nohot"])


def test_merge_scaled_darks(golden):
    g = golden("merge_dark_scaled")
    darks = []
    for i, t in enumerate(g["exposures"]):
        j, sc = orc.select_dark(float(t), [float(x) for x in g["dark_exposures"]], float(g["dark_threshold"]))
        assert j == int(g["dark_sel"][i])
        assert sc == pytest.approx(float(g["dark_scale"][i]), rel=1e-15)
        darks.append(None if j < 0 else (sc * orc.unit_from_u8(g["darks"][j]) if sc != 1.0
                                          else orc.unit_from_u8(g["darks"][j])))
    out = orc.merge(list(g["frames"]), g["exposures"], g["icrf"], g["icrf_diff"], stds=list(g["stds"]),
                    darks=darks, dark_threshold=float(g["dark_threshold"]), median_k=int(g["median_k"]))
    close(out["val"], g["val"])
    close(out["std"], g["std"])


def test_median_matches_scipy():
    scipy_ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(0)
    x = rng.random((9, 13, 3))
    for k in (3, 5):
        ref = scipy_ndimage.median_filter(x, size=(k, k), axes=(0, 1), mode="reflect")
        assert np.array_equal(orc.median_filter_reflect(x, k), ref)


def test_operators(golden):
    g = golden("operators")
    a, b, sa, sb = g["a"], g["b"], g["sa"], g["sb"]
    combos = {"ss": (sa, sb), "sn": (sa, None), "ns": (None, sb), "nn": (None, None)}
    fns = {"add": orc.op_add, "sub": orc.op_sub, "mul": orc.op_mul, "div": orc.op_div, "pow": orc.op_pow}
    for tag, (s1, s2) in combos.items():
        for name, fn in fns.items():
            val, std = fn(a, s1, b, s2)
            close(val, g[f"{name}_{tag}_val"])
            if tag == "nn":
                assert std is None and f"{name}_{tag}_std" not in g
            else:
                close(std, g[f"{name}_{tag}_std"])
    one = np.array([1.0])
    for name, (val, std) in {
        "neg": orc.op_neg(a, sa), "loge": orc.op_log_e(a, sa), "log10": orc.op_log_10(a, sa),
        "rmul": orc.op_mul(a, sa, one * 2.5, None), "adds": orc.op_add(a, sa, one * 1.5, None),
        "subs": orc.op_sub(a, sa, one * 0.125, None), "muls": orc.op_mul(a, sa, one * 3.0, None),
        "divs": orc.op_div(a, sa, one * 4.0, None), "pows": orc.op_pow(a, sa, one * 2, None),
        "sqrt": orc.op_pow(a, sa, one * (1 / 2), None),
    }.items():
        close(val, g[f"{name}_val"])
        close(std, g[f"{name}_std"])
    w, dw = orc.gaussian_weight(a)
    close(w, g["gw_w"])
    close(dw, g["gw_dw"])
    ad, ads, rd, rds = orc.compute_difference(a, sa, g["b2"], g["sb2"], 0.5)
    close(ad, g["cd_abs_val"]); close(ads, g["cd_abs_std"]); close(rd, g["cd_rel_val"]); close(rds, g["cd_rel_std"])
    iv, istd = orc.interpolate(a, sa, g["b2"], g["sb2"], 1.0, 3.0, 1.5)
    close(iv, g["interp_val"]); close(istd, g["interp_std"])
    st = orc.dimension_statistics(g["a_nan"], g["sa_nan"], (0, 1))
    close(st["mean"], g["stat_s_mean"]); close(st["std"], g["stat_s_std"]); close(st["error"], g["stat_s_err"])
    st = orc.dimension_statistics(g["a_nan"], None, (0, 1))
    close(st["mean"], g["stat_n_mean"]); close(st["std"], g["stat_n_std"])
    tv, ts = orc.apply_thresholds(a, sa, [0.4, None, 0.5], [1.0, 0.9, None])
    np.testing.assert_array_equal(tv, g["thr_val"])
    np.testing.assert_array_equal(ts, g["thr_std"])


# ---- Welford mean / std frames (reference: video_processing.welford_algorithm as written) ----
def test_welford_matches_reference(golden):
    g = golden("welford")
    r = orc.welford(list(g["clip"]), None, True)
    assert np.array_equal(r["mean"], g["mean"])
    assert np.array_equal(r["std"], g["std"])
    assert orc.welford(list(g["clip"]), None, False)["std"] is None
    # folding in two batches continues the same state bit for bit
    m1, q1, c1 = orc.welford_state(list(g["clip"][:10]))
    m2, q2, c2 = orc.welford_state(list(g["clip"][10:]), mean=m1, m2=q1, count=c1)
    m, q, c = orc.welford_state(list(g["clip"]))
    assert c2 == c and np.array_equal(m2, m) and np.array_equal(q2, q)


# ---- ICRF-calibration energy function (reference: ICRF_calibration_exposure as written) ----
def test_energy_function_matches_reference(golden):
    g = golden("energy")
    lo, up = int(g["lower"]), int(g["upper"])
    n_valid = 0
    for b, pv in enumerate(g["params"]):
        c = orc.candidate_icrf(g["mean_icrf"], g["pca"], pv)
        assert np.array_equal(c, g["icrfs"][b])
        ok = orc.candidate_valid(c)
        n_valid += ok
        assert ok == np.isfinite(g["energy_plain"][b])
        for sd, key, pkey in ((None, "energy_plain", "pairs_plain"), (g["sd"], "energy_std", "pairs_std")):
            e = orc.energy_function(c, g["dn"], sd, lo, up, g["exposures"])
            np.testing.assert_allclose(e, g[key][b], rtol=RTOL)
            pairs = orc.analyze_linearity_pairs(c[g["dn"]], sd, c[lo], c[up], True, g["exposures"])
            np.testing.assert_allclose(pairs, g[pkey][b], rtol=RTOL, equal_nan=True)
    assert 0 < n_valid < len(g["params"])           # the fixture holds accepted and rejected candidates
    c = g["icrfs"][0]
    np.testing.assert_allclose(orc.analyze_linearity_pairs(c[g["dn"]], None, c[lo], c[up], False, g["exposures"]),
                               g["abs_plain"], rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(orc.analyze_linearity_pairs(c[g["dn"]], g["sd"], c[lo], c[up], False, g["exposures"]),
                               g["abs_std"], rtol=RTOL, equal_nan=True)


def test_config1_shape_identity_icrf():
    """BASELINE configs[0]: 3-frame 256x256x3 uint8 stack, identity ICRF, CPU merge - the plumbing case. With the identity
    ICRF the radiance is the weighted mean of DN/255/t: checked against a direct evaluation, and the index against the DNs."""
    frames, _, t = orc.synthetic_stack(0, 3, 256, 256)
    icrf = np.stack([np.arange(256) / 255.0] * 3, axis=1)
    out = orc.merge(frames, t, icrf)
    assert out["val"].shape == (256, 256, 3) and all(np.array_equal(i, f) for i, f in zip(out["idx"], frames))
    w_lut = orc.gaussian_weight_lut()[0]
    num = sum(w_lut[f] * (f / 255.0) / ti for f, ti in zip(frames, t))
    den = sum(w_lut[f] for f in frames)
    np.testing.assert_allclose(out["val"], num / den, rtol=1e-13)
