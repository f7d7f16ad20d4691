"""GPU tests of the drop-in API (HipMeasurand / ImageSet / ExposureSeries) - modelled on the reference's
tests/unit/test_measurand.py (real arrays + Hypothesis properties, atol 1e-8 there) and pinned to the
operator outputs the reference itself produced (tests/golden/operators.npz)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from hypothesis import given, settings, strategies as st  # noqa: E402

from oracle import hdr_oracle as orc  # noqa: E402

RT = 1e-13


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from camera_linearity_amd.measurand_factory import Measurand
    return Measurand


def close(t, ref, rtol=RT):
    np.testing.assert_allclose(t.cpu().numpy(), ref, rtol=rtol, atol=0)


def test_operators_match_reference_outputs(M, golden):
    g = golden("operators")
    a, b, sa, sb = g["a"], g["b"], g["sa"], g["sb"]
    combos = {"ss": (sa, sb), "sn": (sa, None), "ns": (None, sb), "nn": (None, None)}
    ops = {"add": lambda x, y: x + y, "sub": lambda x, y: x - y, "mul": lambda x, y: x * y,
           "div": lambda x, y: x / y, "pow": lambda x, y: x ** y}
    for tag, (s1, s2) in combos.items():
        A, B = M(a, s1), M(b, s2)
        for name, fn in ops.items():
            r = fn(A, B)
            close(r.val, g[f"{name}_{tag}_val"])
            if tag == "nn":
                assert r.std is None
            else:
                close(r.std, g[f"{name}_{tag}_std"], 1e-12)
    A = M(a, sa)
    for name, r in {"neg": -A, "loge": A.log_e(), "log10": A.log_10(), "rmul": 2.5 * A, "adds": A + 1.5,
                    "subs": A - 0.125, "muls": A * 3.0, "divs": A / 4.0, "pows": A ** 2, "sqrt": A ** (1 / 2)}.items():
        close(r.val, g[f"{name}_val"])
        close(r.std, g[f"{name}_std"], 1e-12)
    An = M(a)
    close((An ** 2).val, g["pows_n_val"])
    assert (An * 3.0).std is None
    w, dw = A.apply_gaussian_weight()
    close(w, g["gw_w"], 1e-14)
    close(dw, g["gw_dw"], 1e-14)
    ex = A.extract([0, 2], axis=-1)
    close(ex.val, g["extract_val"])
    close(ex.std, g["extract_std"])
    B2 = M(g["b2"], g["sb2"])
    ad, rd = A.compute_difference(A, B2, 0.5)
    close(ad.val, g["cd_abs_val"]); close(ad.std, g["cd_abs_std"]); close(rd.val, g["cd_rel_val"]); close(rd.std, g["cd_rel_std"])
    ip = A.interpolate(A, B2, 1.0, 3.0, 1.5)
    close(ip.val, g["interp_val"]); close(ip.std, g["interp_std"])
    st_ = M(g["a_nan"].copy(), g["sa_nan"].copy()).compute_dimension_statistics(axis=(0, 1))
    close(st_["mean"], g["stat_s_mean"], 1e-12); close(st_["std"], g["stat_s_std"], 1e-12); close(st_["error"], g["stat_s_err"], 1e-12)
    st_ = M(g["a_nan"].copy()).compute_dimension_statistics(axis=(0, 1))
    close(st_["mean"], g["stat_n_mean"], 1e-12); close(st_["std"], g["stat_n_std"], 1e-12)
    T = M(a.copy(), sa.copy())
    T.apply_thresholds([0.4, None, 0.5], [1.0, 0.9, None])
    np.testing.assert_array_equal(T.val.cpu().numpy(), g["thr_val"])
    np.testing.assert_array_equal(T.std.cpu().numpy(), g["thr_std"])
    with pytest.raises(ValueError):
        T.apply_thresholds([0.1], None)
    z = A.zeros_like_measurand()
    assert float(z.val.abs().sum()) == 0.0 and z.std.shape == A.std.shape
    with pytest.raises(ValueError):
        M(np.ones((4, 2))) + M(np.ones((3,)))
    with pytest.raises(TypeError):
        A + "x"


# ---- Hypothesis properties (tests/unit/test_measurand.py:170-378)
shapes = st.lists(st.integers(1, 6), min_size=1, max_size=4).map(tuple)


@st.composite
def pair(draw):
    shp = draw(shapes)
    cut = draw(st.integers(0, len(shp) - 1))
    shp2 = tuple(1 if draw(st.booleans()) else s for s in shp[cut:])
    rng = np.random.default_rng(draw(st.integers(0, 2 ** 31)))
    a = rng.random(shp) + 0.05
    b = rng.random(shp2) + 0.05
    sa = 0.1 * a if draw(st.booleans()) else None
    sb = 0.1 * b if draw(st.booleans()) else None
    return a, sa, b, sb


@settings(max_examples=25, deadline=None)
@given(pair())
def test_operator_properties(M, p):
    a, sa, b, sb = p
    A, B = M(a, sa), M(b, sb)
    for name, fn, ofn in (("add", lambda x, y: x + y, orc.op_add), ("sub", lambda x, y: x - y, orc.op_sub),
                          ("mul", lambda x, y: x * y, orc.op_mul), ("div", lambda x, y: x / y, orc.op_div)):
        r = fn(A, B)
        rv, rs = ofn(a, sa, b, sb)
        close(r.val, rv, 1e-14)
        if rs is None:
            assert r.std is None
        else:
            close(r.std, rs, 1e-12)
    # commutativity / identities, atol as in the reference's tests
    np.testing.assert_allclose((A + B).val.cpu().numpy(), (B + A).val.cpu().numpy(), atol=1e-8)
    np.testing.assert_allclose((A * B).val.cpu().numpy(), (B * A).val.cpu().numpy(), atol=1e-8)
    np.testing.assert_allclose((A - A).val.cpu().numpy(), 0, atol=1e-8)
    np.testing.assert_allclose((A / A).val.cpu().numpy(), 1, atol=1e-8)
    np.testing.assert_allclose(((A * B) / B).val.cpu().numpy(), np.broadcast_to(a, np.broadcast_shapes(a.shape, b.shape)), atol=1e-8)


@settings(max_examples=15, deadline=None)
@given(st.integers(1, 4), st.integers(1, 9), st.integers(1, 9), st.integers(0, 2 ** 31), st.booleans())
def test_linearize_property(M, c, h, w, seed, with_std):
    """tests/unit/test_measurand.py:447-467: every output is a member of its channel's ICRF column - and,
    beyond the reference's test, it is exactly ICRF[idx, c] with the shape preserved."""
    rng = np.random.default_rng(seed)
    v = rng.random((h, w, c))
    s = 0.1 * v if with_std else None
    icrf = np.stack([np.linspace(0, 1, 256) ** (k + 1) for k in range(c)], axis=1)
    diff = np.stack([np.gradient(icrf[:, k], 2 / 255) for k in range(c)], axis=1)
    r = M(v, s).linearize(icrf, diff)
    assert tuple(r.val.shape) == (h, w, c)
    rv, rs, idx = orc.linearize(v, s, icrf, diff)
    assert np.array_equal(r.val.cpu().numpy(), rv)
    for k in range(c):
        assert np.isin(r.val[..., k].cpu().numpy(), icrf[:, k]).all()
    if with_std:
        close(r.std, rs, 1e-15)
    else:
        assert r.std is None
    assert np.array_equal(M(v).lut_index().cpu().numpy(), idx)


@pytest.mark.parametrize("use_std", [False, True])
@pytest.mark.parametrize("rng_", [None, (0.1, 0.9)])
def test_channel_histogram_matches_numpy(M, use_std, rng_):
    """compute_channel_histogram (measurand.py:430-469) = np.histogram per channel; counts exact, weighted sums 1e-12."""
    rng = np.random.default_rng(8)
    v = rng.random((64, 48, 3))
    v[3, 4, 0] = np.nan; v[5, 6, 1] = np.inf; v[0, 0, 2] = 0.9; v[1, 1, 2] = 0.1          # non-finite values and exact range edges
    s = 0.05 + rng.random((64, 48, 3))
    s[7, 7, 1] = 0.0
    m = M(v, s)
    got = m.compute_channel_histogram(32, rng_, [0, 2, 1], use_std)
    for c in (0, 1, 2):
        cv = v[..., c]
        mask = np.isfinite(cv)
        w = None
        if use_std:
            mask &= s[..., c] != 0
            w = 1 / s[..., c][mask]
        ref_h, ref_e = np.histogram(cv[mask], bins=32, range=rng_, weights=w)
        np.testing.assert_allclose(got[c][1], ref_e, rtol=0, atol=0)
        if use_std:
            np.testing.assert_allclose(got[c][0], ref_h, rtol=1e-12)
        else:
            assert np.array_equal(got[c][0], ref_h)


def _features(t):
    return {"illumination": "bf", "magnification": "5x", "exposure": float(t), "subject": "s"}


def test_exposure_series_process_hdr_image(golden):
    """The full drop-in path: ImageSets in memory -> ExposureSeries.process_HDR_image -> merged ImageSet,
    with dark-frame filtering and flat-field correction, against the reference-generated vectors."""
    from camera_linearity_amd import settings as gs
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    g = golden("merge_full")
    old = (gs.DARK_THRESHOLD, gs.FF_MID_PERCENTAGE, gs.MEDIAN_FILTER_KERNEL_SIZE)
    gs.configure(DARK_THRESHOLD=float(g["dark_threshold"]), FF_MID_PERCENTAGE=float(g["ff_mid"]),
                 MEDIAN_FILTER_KERNEL_SIZE=int(g["median_k"]))
    try:
        sets = [ImageSet(use_cupy=True, value=g["frames"][i], std=g["stds"][i], features=_features(t)) for i, t in enumerate(g["exposures"])]
        darks = [ImageSet(use_cupy=True, value=g[k], features=dict(_features(e), subject="dark"))
                 for k, e in (("dark16", 0.016), ("dark32", 0.032), ("dark64", 0.064))]
        flat = ImageSet(use_cupy=True, value=g["flat"], std=g["flat_std"], features=dict(_features(0.01), subject="flat"))
        series = ExposureSeries(input_image_sets=sets)
        series.process_HDR_image(g["icrf"], g["icrf_diff"], dark_list=darks, flat_list=[flat])
        val, std = series.merged_image_set.host_arrays()
        np.testing.assert_allclose(val, g["val_ff"], rtol=1e-12)
        np.testing.assert_allclose(std, g["std_ff"], rtol=1e-9)
        assert series.merged_image_set.is_HDR and series.merged_image_set.measurand.backend == "hip"
        # ICRF_diff derived with the reference's gradient convention when not given
        series.process_HDR_image(g["icrf"], dark_list=darks)
        val, std = series.merged_image_set.host_arrays()
        np.testing.assert_allclose(val, g["val"], rtol=1e-12)
        np.testing.assert_allclose(std, g["std"], rtol=1e-9)
        # val-only (no std images)
        sets2 = [ImageSet(use_cupy=True, value=g["frames"][i], features=_features(t)) for i, t in enumerate(g["exposures"])]
        s2 = ExposureSeries(input_image_sets=sets2)
        s2.process_HDR_image(g["icrf"])
        assert s2.merged_image_set.measurand.std is None
        np.testing.assert_allclose(s2.merged_image_set.host_arrays()[0], g["val_nohot"], rtol=1e-12)
        # S and S**2 (exposure_series.py:317-345)
        S, S2 = ExposureSeries(input_image_sets=sets2)._precalculate_sum_of_weights()
        ref = orc.merge(list(g["frames"]), g["exposures"], g["icrf"])
        np.testing.assert_allclose(S.cpu().numpy(), ref["S"], rtol=1e-14)
        # stack-wide linearize and per-image pass-throughs
        lin = series.linearize(g["icrf"], g["icrf_diff"])
        v0, s0, _ = orc.linearize(orc.unit_from_u8(g["frames"][0]), g["stds"][0], g["icrf"], g["icrf_diff"])
        np.testing.assert_array_equal(lin.input_image_sets[0].host_arrays()[0], v0)
        filt = sets[6].bad_pixel_filter(darks[2])
        refv = orc.hot_pixel_filter(orc.unit_from_u8(g["frames"][6]), orc.unit_from_u8(g["dark64"]), float(g["dark_threshold"]), 3)
        np.testing.assert_array_equal(filt.host_arrays()[0], refv)
        scaled = darks[2].scale_to_exposure(0.032)
        assert scaled.features["exposure"] == 0.032 and darks[2].features["exposure"] == 0.064      # deviation I
        np.testing.assert_allclose(scaled.host_arrays()[0], 0.5 * orc.unit_from_u8(g["dark64"]), rtol=1e-15)
    finally:
        gs.configure(DARK_THRESHOLD=old[0], FF_MID_PERCENTAGE=old[1], MEDIAN_FILTER_KERNEL_SIZE=old[2])


def test_exposure_series_scaled_dark_selection(golden):
    from camera_linearity_amd import settings as gs
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    g = golden("merge_dark_scaled")
    old = gs.DARK_THRESHOLD
    gs.configure(DARK_THRESHOLD=float(g["dark_threshold"]))
    try:
        sets = [ImageSet(use_cupy=True, value=g["frames"][i], std=g["stds"][i], features=_features(t)) for i, t in enumerate(g["exposures"])]
        darks = [ImageSet(use_cupy=True, value=g["darks"][i], features=dict(_features(e), subject="dark")) for i, e in enumerate(g["dark_exposures"])]
        series = ExposureSeries(input_image_sets=sets)
        series.process_HDR_image(g["icrf"], g["icrf_diff"], dark_list=darks)
        val, std = series.merged_image_set.host_arrays()
        np.testing.assert_allclose(val, g["val"], rtol=1e-12)
        np.testing.assert_allclose(std, g["std"], rtol=1e-9)
    finally:
        gs.configure(DARK_THRESHOLD=old)


def test_exposure_pairs_and_linearity_stats():
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    frames, stds, t = orc.synthetic_stack(5, 4, 16, 12, with_std=True)
    sets = [ImageSet(use_cupy=True, value=orc.unit_from_u8(f), std=s, features=_features(ti)) for f, s, ti in zip(frames, stds, t)]
    series = ExposureSeries(input_image_sets=sets)
    series.initialize_exposure_pairs()
    assert len(series.exposure_pairs) == 6
    icrf, _ = orc.synthetic_icrf((1.0, 1.0, 1.0))
    series.process_linearity(icrf, linearity_limit=5, use_std=True)
    ab, rel = series.collect_exposure_pair_stats()
    assert ab["means"].shape == (6, 3) and rel["stds"].shape == (6, 3)
    # first pair against the oracle formulas
    lo, hi = icrf[5, 0], icrf[250, 0]
    v0, s0 = orc.apply_thresholds(orc.unit_from_u8(frames[0]), stds[0], [lo] * 3, [hi] * 3)
    v1, s1 = orc.apply_thresholds(orc.unit_from_u8(frames[1]), stds[1], [lo] * 3, [hi] * 3)
    ad, ads, rd, rds = orc.compute_difference(v0, s0, v1, s1, t[0] / t[1])
    ref = orc.dimension_statistics(ad, ads, (0, 1))
    np.testing.assert_allclose(ab["means"][0], ref["mean"], rtol=1e-10)
    np.testing.assert_allclose(ab["stds"][0], ref["std"], rtol=1e-10)
    np.testing.assert_allclose(ab["errors"][0], ref["error"], rtol=1e-10)
    refr = orc.dimension_statistics(rd, rds, (0, 1))
    np.testing.assert_allclose(rel["means"][0], refr["mean"], rtol=1e-10)
    np.testing.assert_allclose(rel["stds"][0], refr["std"], rtol=1e-10)
    # the fused pair kernel against the unfused HIP path (difference images + per-image statistics), with and without std
    from camera_linearity_amd import engine
    for sx, sy in ((s0, s1), (None, None), (s0, None)):
        up = lambda a: None if a is None else torch.as_tensor(a, device="cuda")   # noqa: E731
        fa, fr = engine.pair_statistics(up(v0), up(sx), up(v1), up(sy), t[0] / t[1])
        ad_, ads_, rd_, rds_ = engine.compute_difference(up(v0), up(sx), up(v1), up(sy), t[0] / t[1])
        ua, ur = engine.channel_statistics(ad_, ads_), engine.channel_statistics(rd_, rds_)
        for f, u in ((fa, ua), (fr, ur)):
            for key in ("mean", "std", "error"):
                if u[key] is None:
                    assert f[key] is None
                else:
                    np.testing.assert_allclose(f[key].cpu().numpy(), u[key].cpu().numpy(), rtol=1e-13)


def test_c_abi_example_from_plain_c(tmp_path):
    """The boundary is a C ABI: examples/merge_c_abi.c (C11, gcc, HIP runtime only - no Python, no torch) links
    libhdrmerge.so, merges a stack with hm_merge and checks it against its own host loop."""
    import pathlib
    import shutil
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    root = pathlib.Path(__file__).resolve().parent.parent
    lib = root / "camera_linearity_amd" / "lib"
    exe = tmp_path / "merge_c_abi"
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", str(root / "examples" / "merge_c_abi.c"), f"-I{root / 'include'}",
           "-I/opt/rocm/include", f"-L{lib}", "-lhdrmerge", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}",
           "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI merge OK" in r.stdout
    assert "hipGraph replay of the captured hm_merge: identical bits" in r.stdout      # hipStreamBeginCapture ... hm_merge ... hipGraphLaunch from plain C


def test_out_of_scope_methods_fail_loudly(M):
    """Plotting support the reference computes on the host is not silently done on the CPU here."""
    m = M(np.random.default_rng(0).random((4, 5, 3)))
    with pytest.raises(NotImplementedError):
        m.compute_kernel_density_estimate(10)
    with pytest.raises(NotImplementedError):
        M(np.zeros((2, 2, 5))).compute_channel_histogram(8)


# ------------------------------------------------------------------------------------------------ which path ran
def test_measurand_methods_run_their_hip_kernels(M):
    """Every Measurand method of the statistics / selection group is ONE libhdrmerge kernel (no torch arithmetic behind it):
    the per-symbol call counter of the loaded library must move for exactly the symbol the method documents; shapes the
    kernels do not take raise instead of silently running somewhere else."""
    from camera_linearity_amd import _native as nat
    rng = np.random.default_rng(5)
    a, b = rng.random((6, 7, 3)) + 0.1, rng.random((6, 7, 3)) + 0.1
    A, B = M(a, 0.1 * a), M(b, 0.1 * b)

    def ran(symbol, fn):
        before = dict(nat.lib.calls)
        out = fn()
        moved = {k for k, v in nat.lib.calls.items() if v != before.get(k, 0)}
        assert symbol in moved, (symbol, moved)
        return out
    ex = ran("hm_take_axis", lambda: A.extract([2, 0], axis=-1))
    np.testing.assert_array_equal(ex.val.cpu().numpy(), np.take(a, [2, 0], axis=-1))
    np.testing.assert_array_equal(ex.std.cpu().numpy(), np.take(0.1 * a, [2, 0], axis=-1))
    ex = ran("hm_take_axis", lambda: A.extract([5, 0, 17, -1]))                       # axis=None: the flattened array (np.take)
    np.testing.assert_array_equal(ex.val.cpu().numpy(), np.take(a, [5, 0, 17, -1]))
    ex = ran("hm_take_axis", lambda: A.extract(1, axis=0))
    np.testing.assert_array_equal(ex.val.cpu().numpy(), np.take(a, [1], axis=0))
    with pytest.raises(IndexError):
        A.extract([3], axis=-1)
    with pytest.raises(TypeError):
        A.extract(None, axis=0)
    # more indices than one hm_take_axis launch takes (HM_TAKE_MAX = 16): chunked, np.take has no limit
    many = list(rng.integers(-7, 7, 41))
    ex = ran("hm_take_axis", lambda: A.extract(many, axis=1))                         # outer = 6 > 1: chunks land through strided copies
    np.testing.assert_array_equal(ex.val.cpu().numpy(), np.take(a, many, axis=1))
    np.testing.assert_array_equal(ex.std.cpu().numpy(), np.take(0.1 * a, many, axis=1))
    flat_idx = list(rng.integers(-126, 126, 50))
    np.testing.assert_array_equal(A.extract(flat_idx).val.cpu().numpy(), np.take(a, flat_idx))
    ran("hm_apply_thresholds", lambda: M(a.copy()).apply_thresholds([0.3, None, 0.2], [0.9, 0.8, None]))
    wide = rng.random((5, 9))                                                         # 9 channels: the wide kernel, still HIP
    W9 = M(wide.copy(), 0.1 * wide)
    ran("hm_apply_thresholds", lambda: W9.apply_thresholds([0.2] * 9, [0.7] * 9))
    want = wide.copy()
    want[(want < 0.2) | (want > 0.7)] = np.nan
    np.testing.assert_array_equal(W9.val.cpu().numpy(), want)
    assert np.array_equal(np.isnan(W9.std.cpu().numpy()), np.isnan(want))
    st_ = ran("hm_channel_statistics", lambda: A.compute_dimension_statistics(axis=(0, 1)))
    assert tuple(st_["mean"].shape) == (3,)
    st_all = ran("hm_channel_statistics", lambda: A.compute_dimension_statistics())  # axis=None: every element, 1/std weights
    ref_all = orc.dimension_statistics(a, 0.1 * a, None)
    for key in ("mean", "std", "error"):
        np.testing.assert_allclose(float(st_all[key]), float(ref_all[key]), rtol=1e-12)
    st1 = ran("hm_axis_statistics", lambda: A.compute_dimension_statistics(axis=0))   # any other axis: hm_axis_statistics
    ref1 = orc.dimension_statistics(a, 0.1 * a, 0)
    for key in ("mean", "std", "error"):
        np.testing.assert_allclose(st1[key].cpu().numpy(), ref1[key], rtol=1e-11)
    ran("hm_compute_difference", lambda: A.compute_difference(A, B, 0.5))
    ran("hm_interpolate", lambda: A.interpolate(A, B, 1.0, 3.0, 1.5))
    d_abs, d_rel = ran("hm_compute_difference_bcast", lambda: A.compute_difference(A, M(b[:, :, :1].copy()), 0.5))   # broadcasting operands
    want = orc.compute_difference(a, 0.1 * a, b[:, :, :1], None, 0.5)
    np.testing.assert_allclose(d_abs.val.cpu().numpy(), want[0], rtol=1e-15)
    np.testing.assert_allclose(d_rel.std.cpu().numpy(), want[3], rtol=1e-14)
    ip = ran("hm_interpolate_bcast", lambda: A.interpolate(A, M(b[:1].copy(), 0.2 * b[:1]), 1.0, 3.0, 1.5))
    want = orc.interpolate(a, 0.1 * a, b[:1], 0.2 * b[:1], 1.0, 3.0, 1.5)
    np.testing.assert_allclose(ip.val.cpu().numpy(), want[0], rtol=1e-15)
    np.testing.assert_allclose(ip.std.cpu().numpy(), want[1], rtol=1e-14)
    with pytest.raises(ValueError):
        A.compute_difference(A, M(b[:, :5]), 0.5)                                     # not broadcastable (modules/measurand.py:112)
    ran("hm_binary_op", lambda: A ** B)
    ran("hm_pow_scalar", lambda: A ** 2)
    from camera_linearity_amd import engine
    ran("hm_merge", lambda: engine.merge([torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda")] * 2, [1e-3, 2e-3], orc.synthetic_icrf()[0]))


@pytest.mark.parametrize("p", [2, 0.5, 1, 3, -1, -2, 8, 0, 2.5, -0.75, 9])
def test_pow_scalar_exponent(M, p):
    """Measurand ** plain scalar (hm_pow_scalar: no pow() for 2, 1/2, 1 and small integers) against the reference's formula
    (modules/measurand.py:217-241 evaluated by the oracle with NumPy's pow): value and propagated std to a few ulp; zeros and
    negative values give NumPy's inf / NaN pattern."""
    from camera_linearity_amd import _native as nat
    rng = np.random.default_rng(17)
    a = rng.random((9, 11, 3)) + 0.05
    sa = 0.1 * a
    before = nat.lib.calls["hm_pow_scalar"]
    r = M(a, sa) ** p
    assert nat.lib.calls["hm_pow_scalar"] == before + 1
    rv, rs = orc.op_pow(a, sa, np.array([float(p)]), None)
    close(r.val, rv, 1e-14)
    np.testing.assert_allclose(r.std.cpu().numpy(), rs, rtol=1e-13)
    assert (M(a) ** p).std is None
    odd = np.array([0.0, -1.5, 2.0, np.inf, 0.25, 1.0, -0.0])                 # odd length: the scalar tail of the 2-per-lane kernel
    with np.errstate(all="ignore"):
        ov, os_ = orc.op_pow(odd, 0.1 * np.abs(odd) + 0.01, np.array([float(p)]), None)
    got = M(odd, 0.1 * np.abs(odd) + 0.01) ** p
    gv, gs = got.val.cpu().numpy(), got.std.cpu().numpy()
    assert np.array_equal(np.isnan(gv), np.isnan(ov)) and np.array_equal(np.isinf(gv), np.isinf(ov))
    fin = np.isfinite(ov)
    np.testing.assert_allclose(gv[fin], ov[fin], rtol=1e-14)
    assert np.array_equal(np.isnan(gs), np.isnan(os_)), (p, gs, os_)
    fin = np.isfinite(os_)
    np.testing.assert_allclose(gs[fin], os_[fin], rtol=1e-13)


@pytest.mark.parametrize("use_std", [True, False])
def test_all_pairs_fused_linearity(use_std):
    """ExposureSeries.process_linearity through hm_pairs_statistics (every pair of the series in one launch, frames read once)
    against the oracle's per-pair evaluation (compute_difference + dimension_statistics, modules/exposure_series.py:443-446) and
    against the per-pair HIP kernel; 7 frames with NaNs from the thresholds, 18 pairs (> HM_PAIRS_MAX: two launches)."""
    from camera_linearity_amd import _native as nat, engine
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    n, h, w = 7, 37, 29
    frames, stds, _ = orc.synthetic_stack(8, n, h, w, with_std=True)
    t = 1e-3 * 1.6 ** np.arange(n)                                       # ratios >= 0.1 for |i - j| <= 4: 18 pairs
    sets = [ImageSet(use_cupy=True, value=orc.unit_from_u8(f), std=(s if use_std else None), features=_features(ti)) for f, s, ti in zip(frames, stds, t)]
    series = ExposureSeries(input_image_sets=sets)
    series.initialize_exposure_pairs()
    assert len(series.exposure_pairs) == 18
    icrf, _ = orc.synthetic_icrf((1.0, 1.0, 1.0))
    before = nat.lib.calls["hm_pairs_statistics"]
    series.process_linearity(icrf, linearity_limit=20, use_std=use_std)
    assert nat.lib.calls["hm_pairs_statistics"] == before + 1
    lo, hi = icrf[20, 0], icrf[255 - 20, 0]
    th = [orc.apply_thresholds(orc.unit_from_u8(f), s if use_std else None, [lo] * 3, [hi] * 3) for f, s in zip(frames, stds)]
    assert any(np.isnan(v).any() for v, _ in th)
    up = lambda a: None if a is None else torch.as_tensor(a, device="cuda")   # noqa: E731
    k = 0
    for i in range(n):
        for j in range(n):
            if i >= j or t[i] / t[j] < 0.1:
                continue
            p = series.exposure_pairs[k]
            k += 1
            ad, ads, rd, rds = orc.compute_difference(th[i][0], th[i][1], th[j][0], th[j][1], t[i] / t[j])
            fa, fr = engine.pair_statistics(up(th[i][0]), up(th[i][1]), up(th[j][0]), up(th[j][1]), t[i] / t[j])
            for got, dv, ds, single in ((p.absolute_stats, ad, ads, fa), (p.relative_stats, rd, rds, fr)):
                ref = orc.dimension_statistics(dv, ds, (0, 1))
                for key in ("mean", "std", "error"):
                    if ref[key] is None:
                        assert got[key] is None
                        continue
                    np.testing.assert_allclose(got[key].cpu().numpy(), ref[key], rtol=1e-11)
                    np.testing.assert_allclose(got[key].cpu().numpy(), single[key].cpu().numpy(), rtol=1e-12)
    assert k == 18


@pytest.mark.parametrize("weighted", [False, True])
def test_one_pass_statistics_are_stable(weighted):
    """The one-pass (block-shifted moments + Chan merge) statistics on data that break a naive sum-of-squares: mean 1e6, spread 1e-3
    (relative 1e-9), a quarter of the elements NaN, and a 4096 x 1024 x 3 image so that every level of the merge tree is exercised.
    Against NumPy's two-pass nanmean / nanstd (the reference's formulas, modules/measurand.py:339-349) in float64: std to 1e-8 relative
    (a naive E[x^2] - E[x]^2 in float64 has NO correct digit here), mean to 1e-13 (NumPy's own pairwise sum is no better)."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(23)
    shape = (4096, 1024, 3)
    v = 1.0e6 + 1.0e-3 * rng.standard_normal(shape) + np.array([0.0, 5.0, -3.0])
    v[rng.random(shape) < 0.25] = np.nan
    s = None
    if weighted:
        s = 0.5 + rng.random(shape)
        s[np.isnan(v)] = np.nan
    ref = orc.dimension_statistics(v, s, (0, 1))
    got = engine.channel_statistics(torch.as_tensor(v, device="cuda"), None if s is None else torch.as_tensor(s, device="cuda"))
    np.testing.assert_allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=1e-13)
    np.testing.assert_allclose(got["std"].cpu().numpy(), ref["std"], rtol=1e-8)
    if weighted:
        np.testing.assert_allclose(got["error"].cpu().numpy(), ref["error"], rtol=1e-13)
    # bit-reproducible: a second launch gives the same bits
    again = engine.channel_statistics(torch.as_tensor(v, device="cuda"), None if s is None else torch.as_tensor(s, device="cuda"))
    assert torch.equal(again["std"], got["std"]) and torch.equal(again["mean"], got["mean"])


@pytest.mark.parametrize("with_std", [True, False])
def test_pair_statistics_zero_and_infinite_operands(with_std):
    """hm_pair_statistics / hm_pairs_statistics on operands whose hardware reciprocal / reciprocal-square-root estimates are 0 or
    infinite (y = 0 -> scale = 0; both stds 0 -> variance 0; an infinite value; an infinite std): the wave that meets one redoes the
    element with the guarded forms. Compared with the unfused HIP path (difference images by IEEE divisions + per-image statistics):
    the same NaN / inf pattern per output and 1e-12 where finite; odd sizes, so the partial-chunk path is taken too. A second image
    keeps the special values in its first lanes only, so one workgroup mixes clean and special waves."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(5)
    for shape, special_rows in (((33, 21, 3), None), ((257, 129, 3), 2)):
        x = 0.2 + rng.random(shape)
        y = 0.3 + rng.random(shape)
        sx = 0.01 + 0.01 * rng.random(shape)
        sy = 0.01 + 0.01 * rng.random(shape)
        region = slice(None) if special_rows is None else slice(0, special_rows)
        m = rng.random(shape)
        pick = np.zeros(shape, dtype=bool)
        pick[region] = True
        y[pick & (m < 0.05)] = 0.0
        both0 = pick & (m > 0.05) & (m < 0.10)
        sx[both0] = 0.0
        sy[both0] = 0.0
        x[pick & (m > 0.10) & (m < 0.12)] = np.inf
        sx[pick & (m > 0.12) & (m < 0.14)] = np.inf
        x[pick & (m > 0.14) & (m < 0.20)] = np.nan
        # channel 2 stays clean so that at least one output column is finite
        for a_, fill in ((x, 0.5), (y, 0.6), (sx, 0.01), (sy, 0.02)):
            a_[..., 2] = np.where(np.isfinite(a_[..., 2]) & (a_[..., 2] != 0.0), a_[..., 2], fill)
        up = lambda a: torch.as_tensor(a, device="cuda")   # noqa: E731
        gx, gy = up(x), up(y)
        gsx, gsy = (up(sx), up(sy)) if with_std else (None, None)
        mult = 0.37
        fa, fr = engine.pair_statistics(gx, gsx, gy, gsy, mult)
        ad, ads, rd, rds = engine.compute_difference(gx, gsx, gy, gsy, mult)
        ua, ur = engine.channel_statistics(ad, ads), engine.channel_statistics(rd, rds)
        both = engine.pairs_statistics([gx, gy], None if not with_std else [gsx, gsy], [(0, 1, mult)])
        for f, u, p in ((fa, ua, both[0][0]), (fr, ur, both[0][1])):
            for key in ("mean", "std", "error"):
                if u[key] is None:
                    assert f[key] is None
                    continue
                got, ref, allp = f[key].cpu().numpy(), u[key].cpu().numpy(), p[key].cpu().numpy()
                assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)), (key, got, ref)
                fin = np.isfinite(ref)
                assert fin[2]
                np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-12)
                assert np.array_equal(np.isnan(allp), np.isnan(ref)) and np.array_equal(np.isinf(allp), np.isinf(ref)), (key, allp, ref)
                np.testing.assert_allclose(allp[fin], ref[fin], rtol=1e-12)      # the all-pairs kernel (another partition of the elements)


@pytest.mark.parametrize("with_std", [True, False])
def test_pair_statistics_nan_regions(with_std):
    """Thresholded frames are NaN in whole regions. The pair kernels decide per 64-element chunk and wave whether it is entirely NaN
    (skipped), entirely valid (select-free update) or mixed (general update); all three must give what NumPy's nan-functions give.
    An image whose FIRST rows are NaN (lanes meet valid data late: no shift K yet), with NaN blocks, isolated NaNs, a NaN only in the
    value (its std valid: the weight still counts, measurand.py:342-347) and a NaN only in the std, against the oracle."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(11)
    h, w = 96, 128
    x = 0.2 + rng.random((h, w, 3))
    y = 0.3 + rng.random((h, w, 3))
    sx = 0.01 + 0.01 * rng.random((h, w, 3))
    sy = 0.01 + 0.01 * rng.random((h, w, 3))
    for a_ in (x, sx):
        a_[:7] = np.nan                                  # the image starts with NaN rows
        a_[40:60, 30:90] = np.nan                        # a block
    y[70:80] = np.nan
    sy[70:80] = np.nan
    iso = rng.random((h, w, 3)) < 0.01
    x[iso] = np.nan
    sx[iso] = np.nan
    x[20, 5:9] = np.nan                                  # value NaN, std valid
    sy[25, 5:9] = np.nan                                 # std NaN, value valid
    mult = 0.61
    up = lambda a: torch.as_tensor(a, device="cuda")   # noqa: E731
    args = (up(x), up(sx) if with_std else None, up(y), up(sy) if with_std else None, mult)
    fa, fr = engine.pair_statistics(*args)
    both = engine.pairs_statistics([args[0], args[2]], [args[1], args[3]] if with_std else None, [(0, 1, mult)])
    three = engine.pairs_statistics([args[0], args[2], args[0]], [args[1], args[3], args[1]] if with_std else None,
                                    [(0, 1, mult), (2, 1, mult), (0, 2, 1.0)])           # three frames: the LDS-staged kernel
    ad, ads, rd, rds = orc.compute_difference(x, sx if with_std else None, y, sy if with_std else None, mult)
    for got, p1, p3, dv, ds in ((fa, both[0][0], three[0][0], ad, ads), (fr, both[0][1], three[0][1], rd, rds)):
        ref = orc.dimension_statistics(dv, ds, (0, 1))
        for key in ("mean", "std", "error"):
            if ref[key] is None:
                assert got[key] is None
                continue
            for g_ in (got, p1, p3):
                np.testing.assert_allclose(g_[key].cpu().numpy(), ref[key], rtol=1e-11)


@pytest.mark.parametrize("offset,n", [(0, 4099), (0, 4096), (1, 4097), (1, 4098)])
def test_dense_and_general_kernel_paths_agree(offset, n):
    """hm_binary_op, hm_compute_difference, hm_channel_statistics and hm_pair_statistics pick a two-elements-per-lane kernel for dense,
    16-byte-aligned operands and a general one otherwise. Views that start one float64 into an allocation (8-byte aligned only) and odd
    lengths take the other paths; every path must give NumPy's result (the reference's formulas, measurand.py:106-241, :620-655)."""
    from camera_linearity_amd import engine, _native as nat
    rng = np.random.default_rng(offset * 10 + n)
    host = {k: rng.random(n + 2) + 0.5 for k in ("x", "y")}
    host.update({k: 0.01 + 0.02 * rng.random(n + 2) for k in ("sx", "sy")})
    dev = {k: torch.as_tensor(v, device="cuda")[offset:offset + n] for k, v in host.items()}
    h = {k: v[offset:offset + n] for k, v in host.items()}
    assert all((t_.data_ptr() % 16 == 0) == (offset == 0) for t_ in dev.values())
    x, y, sx, sy = h["x"], h["y"], h["sx"], h["sy"]
    want = {
        nat.HM_OP_ADD: (x + y, np.sqrt(sx ** 2 + sy ** 2)),
        nat.HM_OP_SUB: (x - y, np.sqrt(sx ** 2 + sy ** 2)),
        nat.HM_OP_MUL: (x * y, np.sqrt((x * sy) ** 2 + (y * sx) ** 2)),
        nat.HM_OP_DIV: (x / y, np.sqrt((sx / y) ** 2 + (x * sy / y ** 2) ** 2)),
    }
    for op, (wv, ws) in want.items():
        v, s = engine.elementwise_binary(op, dev["x"], dev["sx"], dev["y"], dev["sy"])
        np.testing.assert_allclose(v.cpu().numpy(), wv, rtol=1e-15)
        np.testing.assert_allclose(s.cpu().numpy(), ws, rtol=1e-14)
        v, s = engine.elementwise_binary(op, dev["x"], None, dev["y"], dev["sy"])          # one std missing -> zeros (:121-124)
        w0 = {nat.HM_OP_ADD: sy, nat.HM_OP_SUB: sy, nat.HM_OP_MUL: np.abs(x * sy), nat.HM_OP_DIV: np.abs(x * sy / y ** 2)}[op]
        np.testing.assert_allclose(s.cpu().numpy(), w0, rtol=1e-14)
        v, s = engine.elementwise_binary(op, dev["x"], None, dev["y"], None)
        assert s is None
        np.testing.assert_allclose(v.cpu().numpy(), wv, rtol=1e-15)
    mult = 0.7
    ad, ads, rd, rds = engine.compute_difference(dev["x"], dev["sx"], dev["y"], dev["sy"], mult)
    oad, oads, ord_, ords = orc.compute_difference(x, sx, y, sy, mult)
    for got, ref in ((ad, oad), (ads, oads), (rd, ord_), (rds, ords)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-14)
    for a_, b_ in ((dev["sx"], dev["sy"]), (dev["sx"], None), (None, None)):
        iv, is_ = engine.interpolate(dev["x"], a_, dev["y"], b_, 0.004, 0.016, 0.007)
        ov, os2 = orc.interpolate(x, None if a_ is None else sx, y, None if b_ is None else sy, 0.004, 0.016, 0.007)
        np.testing.assert_allclose(iv.cpu().numpy(), ov, rtol=1e-15)
        if os2 is None:
            assert is_ is None
        else:
            np.testing.assert_allclose(is_.cpu().numpy(), os2, rtol=1e-14)
    # statistics on (n // 3, 3) views: channel and pair kernels on the same unaligned / odd data
    m = (n // 3) * 3
    as3 = lambda t_: t_[:m].reshape(-1, 3)     # noqa: E731
    fa, fr = engine.pair_statistics(as3(dev["x"]), as3(dev["sx"]), as3(dev["y"]), as3(dev["sy"]), mult)
    for got, dv, ds in ((fa, oad, oads), (fr, ord_, ords)):
        ref = orc.dimension_statistics(dv[:m].reshape(-1, 3), ds[:m].reshape(-1, 3), (0,))
        for key in ("mean", "std", "error"):
            np.testing.assert_allclose(got[key].cpu().numpy(), ref[key], rtol=1e-12)
    st = engine.channel_statistics(as3(dev["x"]), as3(dev["sx"]))
    ref = orc.dimension_statistics(x[:m].reshape(-1, 3), sx[:m].reshape(-1, 3), (0,))
    for key in ("mean", "std", "error"):
        np.testing.assert_allclose(st[key].cpu().numpy(), ref[key], rtol=1e-12)


@pytest.mark.parametrize("with_std", [True, False])
def test_pair_statistics_many_iterations(with_std):
    """The whole-chunk loops of the statistics kernels (prefetching register sets in hm_pair_statistics, the three-stage LDS pipeline
    with one barrier per iteration in hm_pairs_statistics) only run when an image has more than 2 x 768 x 64 elements; the small
    parity cases end in their tail paths. 4 frames of 1024 x 1024 x 3 (+ 100 extra elements so that the tail runs too) with NaNs,
    all 6 pairs: the all-pairs launch, the per-pair launches and NumPy's nan-functions (the oracle) must agree."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(3)
    n_el = 1024 * 1024 + 100
    shape = (n_el, 3)
    vals = [0.2 + rng.random(shape) * (1.0 + 0.3 * i) for i in range(4)]
    stds = [0.01 + 0.02 * rng.random(shape) for _ in range(4)]
    for i, (v, s) in enumerate(zip(vals, stds)):
        m = rng.random(shape) < 0.1
        v[m] = np.nan
        s[m] = np.nan
        v[i * 1000:(i + 1) * 1000 + 5000] = np.nan            # a NaN run (rows of a thresholded image)
        s[i * 1000:(i + 1) * 1000 + 5000] = np.nan
    up = lambda a: torch.as_tensor(a, device="cuda")   # noqa: E731
    gv = [up(v) for v in vals]
    gs = [up(s) for s in stds] if with_std else None
    pairs = [(i, j, 0.4 + 0.1 * (i + j)) for i in range(4) for j in range(i + 1, 4)]
    allp = engine.pairs_statistics(gv, gs, pairs)
    assert len(allp) == 6
    for (i, j, m), got in zip(pairs, allp):
        single = engine.pair_statistics(gv[i], gs[i] if with_std else None, gv[j], gs[j] if with_std else None, m)
        ad, ads, rd, rds = orc.compute_difference(vals[i], stds[i] if with_std else None, vals[j], stds[j] if with_std else None, m)
        for g_, s_, dv, ds in ((got[0], single[0], ad, ads), (got[1], single[1], rd, rds)):
            ref = orc.dimension_statistics(dv, ds, (0,))
            for key in ("mean", "std", "error"):
                if ref[key] is None:
                    assert g_[key] is None and s_[key] is None
                    continue
                np.testing.assert_allclose(g_[key].cpu().numpy(), ref[key], rtol=1e-11)
                np.testing.assert_allclose(s_[key].cpu().numpy(), ref[key], rtol=1e-11)
    # bit-reproducible
    again = engine.pairs_statistics(gv, gs, pairs)
    for a_, b_ in zip(allp, again):
        for h in (0, 1):
            assert torch.equal(a_[h]["mean"], b_[h]["mean"]) and torch.equal(a_[h]["std"], b_[h]["std"])


def test_per_frame_kernels_many_iterations():
    """Linearization (the streaming kernel's register ping-pong loop needs more than 8192 groups of 384 elements per launch), thresholds and
    the Gaussian weight on a 2048 x 2048 x 3 frame (+ a ragged tail), uint8 and float64 input, against the oracle: the small property
    cases end in the kernels' tail / single-iteration paths."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(17)
    n_pix = 2048 * 2048 + 37
    dn = rng.integers(0, 256, (n_pix, 3), dtype=np.uint8)
    sd = 0.004 * (1 + rng.random((n_pix, 3)))
    icrf, diff = orc.synthetic_icrf()
    g_dn, g_sd = torch.as_tensor(dn, device="cuda"), torch.as_tensor(sd, device="cuda")
    # uint8 input with std, per-channel tables
    v, s = engine.linearize(g_dn, g_sd, icrf, diff)
    ov, os_, _ = orc.linearize(dn, sd, icrf, diff)
    assert np.array_equal(v.cpu().numpy(), ov)
    np.testing.assert_allclose(s.cpu().numpy(), os_, rtol=1e-15)
    # float64 input (with .5 ties and values past 1.0), index returned
    x = dn.astype(np.float64) / 255.0
    x[::7] += 0.5 / 255.0
    x[::11] += 1.0
    v, s, idx = engine.linearize(torch.as_tensor(x, device="cuda"), g_sd, icrf, diff, return_index=True)
    ov, os_, oidx = orc.linearize(x, sd, icrf, diff)
    assert np.array_equal(idx.cpu().numpy(), oidx)
    assert np.array_equal(v.cpu().numpy(), ov)
    np.testing.assert_allclose(s.cpu().numpy(), os_, rtol=1e-15)
    # thresholds in place
    lo, hi = [0.1, 0.2, 0.3], [0.8, 0.7, 0.9]
    tv, ts = torch.as_tensor(ov, device="cuda").clone(), torch.as_tensor(os_, device="cuda").clone()
    engine.apply_thresholds_(tv, ts, lo, hi)
    rv, rs = orc.apply_thresholds(ov, os_, lo, hi)
    np.testing.assert_array_equal(tv.cpu().numpy(), rv)
    np.testing.assert_array_equal(ts.cpu().numpy(), rs)
    # Gaussian weight of the uint8 frame (LUT) and of the float64 frame (analytic)
    w, dw = engine.gaussian_weight(g_dn)
    ow, odw = orc.gaussian_weight(dn.astype(np.float64) / 255.0)
    np.testing.assert_allclose(w.cpu().numpy(), ow, rtol=1e-15)
    np.testing.assert_allclose(dw.cpu().numpy(), odw, rtol=1e-14, atol=1e-300)
    w, dw = engine.gaussian_weight(torch.as_tensor(x, device="cuda"))
    ow, odw = orc.gaussian_weight(x)
    np.testing.assert_allclose(w.cpu().numpy(), ow, rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(dw.cpu().numpy(), odw, rtol=1e-13, atol=1e-300)


def test_debug_copy_probe_copies():
    """hm_debug_copy_probe (the copy rate bench.py prints beside the roofline) really copies: whole 4 KB wave chunks, a partial last chunk,
    a buffer smaller than one chunk."""
    from camera_linearity_amd import _native as nat
    st = torch.cuda.current_stream().cuda_stream
    for n in (2, 510, 512 * 4 * 7, 512 * 4 * 4096 + 1022, 3_000_002):
        src = torch.rand(n, dtype=torch.float64, device="cuda")
        dst = torch.zeros(n + 2, dtype=torch.float64, device="cuda")
        nat.check(nat.lib.hm_debug_copy_probe(src.data_ptr(), dst.data_ptr(), n * 8, st), "copy probe")
        torch.cuda.synchronize()
        assert torch.equal(dst[:n], src) and float(dst[n]) == 0.0 and float(dst[n + 1]) == 0.0


@pytest.mark.parametrize("use_std", [True, False])
@pytest.mark.parametrize("h,w,n", [(37, 29, 5), (640, 512, 5), (320, 512, 7)])
def test_process_linearity_thresholds_in_place(use_std, h, w, n):
    """process_linearity leaves the series' image sets thresholded (modules/exposure_series.py:437-441 calls apply_thresholds in place) and
    compares the thresholded frames. Here the thresholds ride on the all-pairs launch: whole iterations are thresholded by the LDS loader,
    the ragged rest by hm_apply_thresholds' kernel (small image: everything by the latter). Afterwards every image set must hold exactly
    what the oracle's apply_thresholds gives (same NaNs, untouched values bit for bit), and the statistics must be those of the
    thresholded frames; a second call changes nothing."""
    from camera_linearity_amd import _native as nat
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    frames, stds, _ = orc.synthetic_stack(21, n, h, w, with_std=True)
    t = 1e-3 * (1.7 if n == 5 else 1.6) ** np.arange(n)                       # n = 7: 18 pairs, i.e. two launches (HM_PAIRS_MAX = 16)
    sets = [ImageSet(use_cupy=True, value=orc.unit_from_u8(f), std=(s if use_std else None), features=_features(ti)) for f, s, ti in zip(frames, stds, t)]
    series = ExposureSeries(input_image_sets=sets)
    series.initialize_exposure_pairs()
    icrf, _ = orc.synthetic_icrf((1.0, 1.0, 1.0))
    before = nat.lib.calls["hm_pairs_statistics"]
    thr_calls = nat.lib.calls["hm_apply_thresholds"]
    series.process_linearity(icrf, linearity_limit=25, use_std=use_std)
    assert nat.lib.calls["hm_pairs_statistics"] == before + 1
    assert nat.lib.calls["hm_apply_thresholds"] == thr_calls                  # no separate thresholding pass from the host side
    lo, hi = icrf[25, 0], icrf[255 - 25, 0]
    th = [orc.apply_thresholds(orc.unit_from_u8(f), s if use_std else None, [lo] * 3, [hi] * 3) for f, s in zip(frames, stds)]
    assert any(np.isnan(v).any() for v, _ in th) and any(np.isfinite(v).any() for v, _ in th)
    for s_, (rv, rs) in zip(sets, th):
        gv, gs = s_.measurand.to_numpy() if hasattr(s_.measurand, "to_numpy") else (s_.measurand.val.cpu().numpy(), None)
        np.testing.assert_array_equal(np.asarray(gv), rv)
        if use_std:
            np.testing.assert_array_equal(np.asarray(gs), rs)
    k = 0
    for i in range(n):
        for j in range(n):
            if i >= j or t[i] / t[j] < 0.1:
                continue
            p = series.exposure_pairs[k]
            k += 1
            ad, ads, rd, rds = orc.compute_difference(th[i][0], th[i][1], th[j][0], th[j][1], t[i] / t[j])
            for got, dv, ds in ((p.absolute_stats, ad, ads), (p.relative_stats, rd, rds)):
                ref = orc.dimension_statistics(dv, ds, (0, 1))
                for key in ("mean", "std", "error"):
                    if ref[key] is None:
                        assert got[key] is None
                        continue
                    np.testing.assert_allclose(got[key].cpu().numpy(), ref[key], rtol=1e-11)
    assert k == len(series.exposure_pairs)
    first = [(p.absolute_stats["mean"].clone(), p.relative_stats["std"].clone()) for p in series.exposure_pairs]
    series.process_linearity(icrf, linearity_limit=25, use_std=use_std)            # idempotent
    for p, (m_, s__) in zip(series.exposure_pairs, first):
        assert np.array_equal(p.absolute_stats["mean"].cpu().numpy(), m_.cpu().numpy(), equal_nan=True)
        assert np.array_equal(p.relative_stats["std"].cpu().numpy(), s__.cpu().numpy(), equal_nan=True)


# ------------------------------------------------------------------------------------------------ statistics over any axis
@pytest.mark.parametrize("shape,axis", [((40, 50, 3), 0), ((40, 50, 3), 1), ((40, 50, 3), 2), ((40, 50, 3), -1), ((40, 50, 3), (0, 2)),
                                        ((40, 50, 3), (1, 2)), ((40, 50, 3), (0, 1, 2)), ((5, 2000, 3), 1), ((3000, 7), 0), ((3000, 7), 1),
                                        ((2, 3, 4, 5, 6), (1, 3)), ((70000,), 0), ((1, 100000, 2), 1), ((300, 40), 0), ((9, 6), (0,)),
                                        ((64, 64, 8), (0, 1)), ((33, 17, 20), 1),
                                        # two separate groups of reduced axes (hm_axis_statistics2, no layout copy) and three (layout copy)
                                        ((300, 200, 3), (0, 2)), ((6, 5, 40, 3), (0, 2)), ((6, 5, 40, 3), (1, 3)), ((4, 3, 5, 2, 7), (0, 1, 3)),
                                        ((4, 3, 5, 2, 7), (0, 3, 4)), ((2000, 4, 2), (0, 2)), ((3, 70, 1, 9), (0, 3)), ((4, 3, 5, 2, 7), (0, 2, 4)),
                                        # the second group longer than the first (stage 1 then reduces IT), few outputs with many states each (tree fold)
                                        ((3, 4, 5000), (0, 2)), ((64, 8, 4096), (0, 2)), ((2, 3, 500, 2), (0, 2)), ((300, 2, 4000), (0, 2)), ((4096, 2, 3), (0, 2))])
@pytest.mark.parametrize("weighted", [False, True])
def test_dimension_statistics_any_axis(M, shape, axis, weighted):
    """compute_dimension_statistics(axis) for single axes, adjacent and non-adjacent axis tuples, long and short axes, few and many
    outputs (both kernels of hm_axis_statistics, with and without the segment split) against the oracle's NumPy nan-reductions
    (modules/measurand.py:318-350), NaNs in the values and in the stds. rtol 1e-11 (one-pass moments against two NumPy passes)."""
    rng = np.random.default_rng([int(np.prod(shape)), len(shape), sum(np.atleast_1d(axis).tolist()) + 7, int(weighted)])
    a = rng.random(shape) + 0.25
    s_ = 0.05 + 0.1 * rng.random(shape) if weighted else None
    a[rng.random(shape) < 0.05] = np.nan
    if weighted:
        s_[rng.random(shape) < 0.03] = np.nan
    ax = (axis,) if isinstance(axis, int) else tuple(axis)
    leading = sorted(x % a.ndim for x in ax) == list(range(len(ax)))
    with np.errstate(all="ignore"):
        if not weighted or leading:
            ref = orc.dimension_statistics(a, s_, axis)
        else:
            # the reference's weighted branch subtracts the REDUCED mean from the values (modules/measurand.py:344): that only broadcasts
            # when the reduced axes lead; for the other axes it raises (or, on coinciding sizes, mis-broadcasts). The kernel computes what
            # the formula means - the mean kept along the reduced axes - which is the same expression with keepdims:
            w = 1 / s_
            sw = np.nansum(w, axis=ax, keepdims=True)
            mean = np.nansum(a * w, axis=ax, keepdims=True) / sw
            sd = np.sqrt(np.nansum(w * (a - mean) ** 2, axis=ax, keepdims=True) / sw)
            ref = dict(mean=np.squeeze(mean, axis=ax), std=np.squeeze(sd, axis=ax), error=np.nanmean(s_, axis=ax))
    got = M(a.copy(), None if s_ is None else s_.copy()).compute_dimension_statistics(axis=axis)
    for key in ("mean", "std") + (("error",) if weighted else ()):
        g_ = got[key].cpu().numpy()
        assert g_.shape == np.asarray(ref[key]).shape, (key, g_.shape, np.asarray(ref[key]).shape)
        # (atol: a line with ONE counted element has std exactly 0 in the two-pass form and a rounding residue of the mean, ~1e-16, here)
        np.testing.assert_allclose(g_, ref[key], rtol=1e-11, atol=1e-13, equal_nan=True)
    if not weighted:
        assert got["error"] is None


def test_dimension_statistics_all_nan_column_and_errors(M):
    a = np.ones((6, 5, 3))
    a[:, 2, 1] = np.nan                                              # a whole reduction line of NaNs: NumPy gives NaN (with a warning)
    got = M(a).compute_dimension_statistics(axis=0)
    m = got["mean"].cpu().numpy()
    assert np.isnan(m[2, 1]) and np.all(m[np.isfinite(m)] == 1.0) and np.isnan(got["std"].cpu().numpy()[2, 1])
    with pytest.raises(ValueError):
        M(a).compute_dimension_statistics(axis=3)


def test_device_and_host_builds_agree_on_random_cases():
    """tools/fuzz_backends.py for 12 seconds (~10 000 random cases over merge, operators, statistics, thresholds / differences, linearize,
    corrections, histogram / extract): the HIP library against the independent host build of the same ABI - bit-exact where the operation
    sequence is shared (uint8 merges with every correction, gathers, filters), to rounding where two math libraries meet."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_backends.py"), "--seconds", "12", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " 0 failures" in r.stdout


@pytest.mark.parametrize("weighted", [False, True])
def test_statistics_with_infinite_values_answer_like_numpy(M, weighted):
    """Infinite values (a relative difference against y = 0 is one): np.nanmean keeps them (+-inf by sign, NaN for both signs), np.nanstd
    turns NaN, the weighted formulas of modules/measurand.py:342-346 give inf or 0 (their nansum skips the inf - inf terms). The one-pass
    moments keep infinite terms out (inf - shift would be NaN) and carry their sum separately; every statistics kernel must answer as
    the NumPy oracle does: all-but-last axes (hm_channel_statistics), other axes (hm_axis_statistics), whole array, and the fused pair
    kernels against the unfused path."""
    from camera_linearity_amd import engine
    rng = np.random.default_rng(42)
    shape = (37, 29, 4)
    x = rng.normal(size=shape) + 3.0
    x[rng.random(shape) < 0.1] = np.nan
    x[3, 5, 0] = np.inf                        # channel 0: +inf only
    x[7, 1, 1] = -np.inf                       # channel 1: -inf only
    x[0, 0, 2] = np.inf; x[36, 28, 2] = -np.inf; x[5, 5, 2] = np.inf      # channel 2: both signs
    #                                            channel 3: clean
    s = 0.01 + 0.05 * rng.random(shape) if weighted else None
    m = M(x, s)
    with np.errstate(all="ignore"):
        for axis in ((0, 1), 0, 1, 2, None, (1, 2)):
            if weighted and axis not in ((0, 1), None, 0):
                continue                                   # (weighted statistics on other axes: the reference's formula does not broadcast)
            ref = orc.dimension_statistics(x, s, axis)
            got = m.compute_dimension_statistics(axis)
            for key in ("mean", "std"):
                g, r = got[key].cpu().numpy(), np.asarray(ref[key])
                assert np.array_equal(np.isnan(g), np.isnan(r)), (axis, key, g, r)
                assert np.array_equal(np.isinf(g), np.isinf(r)) and np.array_equal(g[np.isinf(g)], r[np.isinf(r)]), (axis, key, g, r)
                fin = np.isfinite(r)
                np.testing.assert_allclose(g[fin], r[fin], rtol=1e-11)
    # a line of ONLY infinite values: mean inf, weighted std 0 (every term skipped), unweighted NaN
    y = np.full((5, 3), np.inf)
    y[:, 2] = 1.5
    sy = np.full((5, 3), 0.1) if weighted else None
    with np.errstate(all="ignore"):
        ref = orc.dimension_statistics(y, sy, 0)
    got = M(y, sy).compute_dimension_statistics(0)
    for key in ("mean", "std"):
        g, r = got[key].cpu().numpy(), np.asarray(ref[key])
        assert np.array_equal(g, r, equal_nan=True), (key, g, r)
    # the fused pair kernels: y = 0 makes the relative difference infinite
    a = 0.2 + rng.random((64, 48, 3)); b = 0.3 + rng.random((64, 48, 3))
    b[10, 10, 0] = 0.0; b[0, 0, 1] = 0.0; b[63, 47, 1] = 0.0          # +inf (channel 1: in the very first and the very last element)
    a[5, 5, 2] = -0.3; b[5, 5, 2] = 0.0; b[6, 6, 2] = 0.0               # channel 2: -inf and +inf -> mean NaN
    sa = 0.01 + 0.01 * rng.random(a.shape) if weighted else None
    sb = 0.01 + 0.01 * rng.random(a.shape) if weighted else None
    up = lambda v: None if v is None else torch.as_tensor(v, device="cuda")   # noqa: E731
    with np.errstate(all="ignore"):
        ad, ads, rd, rds = orc.compute_difference(a, sa, b, sb, 0.5)
        refs = (orc.dimension_statistics(ad, ads, (0, 1)), orc.dimension_statistics(rd, rds, (0, 1)))
    one = engine.pair_statistics(up(a), up(sa), up(b), up(sb), 0.5)
    many = engine.pairs_statistics([up(a), up(b)], None if not weighted else [up(sa), up(sb)], [(0, 1, 0.5)])[0]
    for got2 in (one, many):
        for g_, r_ in zip(got2, refs):
            for key in ("mean", "std"):
                g, r = g_[key].cpu().numpy(), np.asarray(r_[key])
                assert np.array_equal(np.isnan(g), np.isnan(r)) and np.array_equal(np.isinf(g), np.isinf(r)), (key, g, r)
                fin = np.isfinite(r)
                np.testing.assert_allclose(g[fin], r[fin], rtol=1e-9)
                assert np.array_equal(g[~fin], r[~fin], equal_nan=True), (key, g, r)


def test_statistics_zero_weight_first_element_and_no_spread(M):
    """Two findings of tools/fuzz_backends.py with special values. A weighted line whose FIRST element has std = inf (weight 0) and whose
    other element carries all the weight: the moments' shift must come from an element that counts (a shift 3 units from the weighted mean
    gave std 3.5e-8 where NumPy has 0). And no spread at all must give 0, not sqrt(-1e-13) = NaN."""
    x = np.array([[-80.964321, -84.078347], [-42.027734, -44.910037], [5.0, 5.0]])
    s = np.array([[np.inf, 0.01097], [np.inf, 0.010663], [0.1, 0.1]])
    x, s = np.ascontiguousarray(x.T), np.ascontiguousarray(s.T)          # lines along axis 0 (the reference's weighted formula broadcasts only there)
    got = M(x, s).compute_dimension_statistics(0)
    with np.errstate(all="ignore"):
        ref = orc.dimension_statistics(x, s, 0)
    np.testing.assert_allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=1e-14)
    np.testing.assert_allclose(got["std"].cpu().numpy(), ref["std"], rtol=0, atol=1e-13)
    # the same through the all-but-last-axes kernel and with many elements per lane
    rng = np.random.default_rng(3)
    big = np.full((4000, 3), 7.25); sb = np.full((4000, 3), np.inf)
    big[:, 1] = rng.normal(size=4000); sb[:, 1] = 0.1
    sb[1234, 0] = 0.5; sb[0, 2] = np.inf; sb[3999, 2] = 0.25; big[0, 2] = -1e6          # channel 2: a far outlier of weight 0 comes first
    got = M(big, sb).compute_dimension_statistics(0)
    with np.errstate(all="ignore"):
        ref = orc.dimension_statistics(big, sb, 0)
    np.testing.assert_allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=1e-12)
    np.testing.assert_allclose(got["std"].cpu().numpy(), ref["std"], rtol=1e-10, atol=1e-12)


def test_weighted_statistics_with_a_negligible_outlier_first(M):
    """Weighted statistics whose FIRST element per line is a huge value with a negligible weight (a relative difference against y ~ 0: 1.4e6 with
    weight 4e-12 beside values of 0.05 with weight 54 - the case tools/fuzz_backends.py found): as the moments' shift it cost 2.8e-6 on the std;
    the channel / axis statistics kernels now take the heavier of a lane's first two elements as the shift. Against the NumPy oracle at 1e-10."""
    rng = np.random.default_rng(8)
    n, c = 5000, 3
    x = 0.05 * rng.normal(size=(n, c))
    s = np.full((n, c), 1 / 54.0) * (1 + 0.1 * rng.random((n, c)))
    x[0, :] = [1.4e6, -3.0e5, 8.0e4]                       # the first element of every column
    s[0, :] = [2.6e11, 1.0e10, 5.0e9]
    x[1234, 1] = 2.0e6; s[1234, 1] = 1.0e12               # and one in the middle of a column
    with np.errstate(all="ignore"):
        ref = orc.dimension_statistics(x, s, 0)
    got = M(x, s).compute_dimension_statistics(0)
    np.testing.assert_allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(got["std"].cpu().numpy(), ref["std"], rtol=1e-10)
    # the same columns as rows of a 3-D array: axis (0, 1) -> hm_channel_statistics, axis 1 -> hm_axis_statistics (other kernels, same rule)
    x3, s3 = x.reshape(50, 100, c), s.reshape(50, 100, c)
    with np.errstate(all="ignore"):
        ref = orc.dimension_statistics(x3, s3, (0, 1))
    got = M(x3, s3).compute_dimension_statistics((0, 1))
    np.testing.assert_allclose(got["std"].cpu().numpy(), ref["std"], rtol=1e-10)
