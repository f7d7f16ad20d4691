"""The host backend - Measurand(use_cupy=False), the reference's NumpyMeasurand slot (modules/measurand_factory.py:10-14,
modules/measurand.py:684-714) - backed by libhdrmerge_host.so, the HOST build of the C ABI (csrc_host/hm_host.cpp: plain C++,
nothing from oracle/). CPU tests: they run without a GPU and never touch the HIP library's compute entry points.

Pinned like the HIP path: against the reference-generated golden vectors (tests/golden/*.npz) and the NumPy oracle, same
tolerances (index exact; val 1e-12 uint8 / 1e-11 float64 frames; std 1e-9; statistics 1e-11). BASELINE.json configs[0] - "3-frame
256x256x3 uint8 synthetic stack, identity ICRF, NumPy Measurand CPU merge (plumbing, no GPU)" - runs literally here."""
import pathlib

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import hdr_oracle as orc  # noqa: E402

VAL_RTOL, STD_RTOL, F64_RTOL = 1e-12, 1e-9, 1e-11


def _features(t):
    return {"illumination": "bf", "magnification": "5x", "exposure": float(t), "subject": "s"}


def H(val=None, std=None):
    from camera_linearity_amd.measurand_factory import Measurand
    return Measurand(val, std, use_cupy=False)


@pytest.fixture()
def heng():
    """engine.<fn> under nat.host_mode(), as HostMeasurand calls it."""
    from camera_linearity_amd.measurand import _HOST_ENGINE
    return _HOST_ENGINE


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_config1_numpy_measurand_cpu_merge():
    """BASELINE.json configs[0]: 3 x 256 x 256 x 3 uint8, identity ICRF, host ImageSets -> ExposureSeries.process_HDR_image, no GPU."""
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    frames, _, t = orc.synthetic_stack(1, 3, 256, 256)
    icrf, _ = orc.synthetic_icrf((1.0, 1.0, 1.0))
    sets = [ImageSet(value=f, features=_features(ti)) for f, ti in zip(frames, t)]              # use_cupy defaults to False (image_set.py:29)
    assert all(s.use_cupy is False and s.measurand.backend == "numpy" for s in sets)
    series = ExposureSeries(input_image_sets=sets)
    assert series.use_cupy is False
    series.process_HDR_image(icrf, use_std=False)
    m = series.merged_image_set.measurand
    assert m.backend == "numpy" and isinstance(m.val, np.ndarray) and m.std is None and series.merged_image_set.is_HDR
    ref = orc.merge(frames, t, icrf)
    np.testing.assert_allclose(m.val, ref["val"], rtol=VAL_RTOL)


@pytest.mark.parametrize("name", ["merge_identity", "merge_std", "merge_ramp"])
def test_host_merge_golden_u8(heng, golden, name):
    g = golden(name)
    stds = [T(s) for s in g["stds"]] if "stds" in g else None
    out = heng.merge([T(f) for f in g["frames"]], g["exposures"], g["icrf"], g["icrf_diff"], stds, want_sum_w=True)
    np.testing.assert_allclose(out["sum_w"].numpy(), g["S"], rtol=8e-15)
    np.testing.assert_allclose(out["val"].numpy(), g["val"], rtol=VAL_RTOL)
    if stds is not None:
        np.testing.assert_allclose(out["std"].numpy(), g["std"], rtol=STD_RTOL)


def test_host_merge_golden_float_frames_and_index(heng, golden):
    g = golden("merge_float")
    out = heng.merge([T(f) for f in g["frames_f64"]], g["exposures"], g["icrf"], g["icrf_diff"], [T(s) for s in g["stds"]])
    np.testing.assert_allclose(out["val"].numpy(), g["val"], rtol=F64_RTOL)
    np.testing.assert_allclose(out["std"].numpy(), g["std"], rtol=STD_RTOL)
    for i in range(3):                                                   # the uint8 LUT index incl. .5 ties, > 1.0 wrap and negatives: bit-exact
        m = H(g["frames_f64"][i].copy())
        assert np.array_equal(m.lut_index(), g["idx"][i])
    lin = H(g["frames_f64"][0].copy(), g["stds"][0].copy()).linearize(g["icrf"], g["icrf_diff"])
    assert np.array_equal(lin.val, g["lin0_val"])
    np.testing.assert_allclose(lin.std, g["lin0_std"], rtol=1e-15)
    w, dw = H(g["frames_f64"][0].copy()).apply_gaussian_weight()
    np.testing.assert_allclose(w, g["w0"], rtol=1e-14)
    np.testing.assert_allclose(dw, g["dw0"], rtol=1e-14)


def test_host_exposure_series_full_corrections(golden):
    """The reference-generated full case (std, dark maps with the exposure rule, hot-pixel medians, flat field) through the host ImageSet /
    ExposureSeries classes."""
    from camera_linearity_amd import settings as gs
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    g = golden("merge_full")
    old = (gs.DARK_THRESHOLD, gs.FF_MID_PERCENTAGE, gs.MEDIAN_FILTER_KERNEL_SIZE)
    gs.configure(DARK_THRESHOLD=float(g["dark_threshold"]), FF_MID_PERCENTAGE=float(g["ff_mid"]), MEDIAN_FILTER_KERNEL_SIZE=int(g["median_k"]))
    try:
        sets = [ImageSet(value=g["frames"][i], std=g["stds"][i], features=_features(t)) for i, t in enumerate(g["exposures"])]
        darks = [ImageSet(value=g[k], features=dict(_features(e), subject="dark")) for k, e in (("dark16", 0.016), ("dark32", 0.032), ("dark64", 0.064))]
        flat = ImageSet(value=g["flat"], std=g["flat_std"], features=dict(_features(0.01), subject="flat"))
        series = ExposureSeries(input_image_sets=sets)
        series.process_HDR_image(g["icrf"], g["icrf_diff"], dark_list=darks, flat_list=[flat])
        val, std = series.merged_image_set.host_arrays()
        np.testing.assert_allclose(val, g["val_ff"], rtol=VAL_RTOL)
        np.testing.assert_allclose(std, g["std_ff"], rtol=STD_RTOL)
        assert series.merged_image_set.measurand.backend == "numpy"
        series.process_HDR_image(g["icrf"], dark_list=darks)              # ICRF_diff derived with the reference's gradient convention
        val, std = series.merged_image_set.host_arrays()
        np.testing.assert_allclose(val, g["val"], rtol=VAL_RTOL)
        np.testing.assert_allclose(std, g["std"], rtol=STD_RTOL)
        S, S2 = ExposureSeries(input_image_sets=[ImageSet(value=f, features=_features(t)) for f, t in zip(g["frames"], g["exposures"])])._precalculate_sum_of_weights()
        ref = orc.merge(list(g["frames"]), g["exposures"], g["icrf"])
        assert isinstance(S, np.ndarray)
        np.testing.assert_allclose(S, ref["S"], rtol=1e-14)
        np.testing.assert_allclose(S2, ref["S"] ** 2, rtol=1e-14)
        filt = sets[6].bad_pixel_filter(darks[2])
        refv = orc.hot_pixel_filter(orc.unit_from_u8(g["frames"][6]), orc.unit_from_u8(g["dark64"]), float(g["dark_threshold"]), 3)
        np.testing.assert_array_equal(filt.measurand.val, refv)
        ffc = series.merged_image_set.flat_field_correction(flat)
        np.testing.assert_allclose(ffc.measurand.val, g["val_ff"], rtol=VAL_RTOL)
    finally:
        gs.configure(DARK_THRESHOLD=old[0], FF_MID_PERCENTAGE=old[1], MEDIAN_FILTER_KERNEL_SIZE=old[2])


def test_host_operators_match_reference_outputs(golden):
    """Row 10 against operators.npz (the reference classes' own outputs): the host build of hm_binary_op / hm_unary_op / hm_pow_scalar."""
    g = golden("operators")
    a, b, sa, sb = g["a"], g["b"], g["sa"], g["sb"]
    combos = {"ss": (sa, sb), "sn": (sa, None), "ns": (None, sb), "nn": (None, None)}
    for tag, (s1, s2) in combos.items():
        A, B = H(a.copy(), None if s1 is None else s1.copy()), H(b.copy(), None if s2 is None else s2.copy())
        for opname, fn in (("add", lambda x, y: x + y), ("sub", lambda x, y: x - y), ("mul", lambda x, y: x * y), ("div", lambda x, y: x / y),
                           ("pow", lambda x, y: x ** y)):
            r = fn(A, B)
            np.testing.assert_allclose(r.val, g[f"{opname}_{tag}_val"], rtol=1e-13)
            if f"{opname}_{tag}_std" in g:
                np.testing.assert_allclose(r.std, g[f"{opname}_{tag}_std"], rtol=1e-12)
            else:
                assert r.std is None
    A = H(a.copy(), sa.copy())
    for opname, r in (("neg", -A), ("loge", A.log_e()), ("log10", A.log_10()), ("rmul", 2.5 * A), ("adds", A + 1.5), ("subs", A - 0.125),
                      ("muls", A * 3.0), ("divs", A / 4.0), ("pows", A ** 2), ("sqrt", A ** (1 / 2))):
        np.testing.assert_allclose(r.val, g[f"{opname}_val"], rtol=1e-13)
        np.testing.assert_allclose(r.std, g[f"{opname}_std"], rtol=1e-12)


@pytest.mark.parametrize("axis", [None, 0, 1, (0, 1), (0, 2), -1])
@pytest.mark.parametrize("weighted", [False, True])
def test_host_dimension_statistics(axis, weighted):
    rng = np.random.default_rng(11)
    a = rng.random((13, 17, 3)) + 0.25
    s_ = 0.05 + 0.1 * rng.random(a.shape) if weighted else None
    a[rng.random(a.shape) < 0.05] = np.nan
    if weighted:
        s_[rng.random(a.shape) < 0.03] = np.nan
    ax = None if axis is None else ((axis,) if isinstance(axis, int) else tuple(axis))
    leading = ax is None or sorted(x % a.ndim for x in ax) == list(range(len(ax)))
    with np.errstate(all="ignore"):
        if not weighted or leading:
            ref = orc.dimension_statistics(a, s_, axis)
        else:                                                            # the reference's formula with the mean kept along the reduced axes (see test_gpu_api)
            w = 1 / s_
            sw = np.nansum(w, axis=ax, keepdims=True)
            mean = np.nansum(a * w, axis=ax, keepdims=True) / sw
            sd = np.sqrt(np.nansum(w * (a - mean) ** 2, axis=ax, keepdims=True) / sw)
            ref = dict(mean=np.squeeze(mean, axis=ax), std=np.squeeze(sd, axis=ax), error=np.nanmean(s_, axis=ax))
    got = H(a.copy(), None if s_ is None else s_.copy()).compute_dimension_statistics(axis=axis)
    for key in ("mean", "std") + (("error",) if weighted else ()):
        assert isinstance(got[key], np.ndarray) and got[key].shape == np.asarray(ref[key]).shape
        np.testing.assert_allclose(got[key], ref[key], rtol=1e-11, atol=1e-13, equal_nan=True)


def test_host_process_linearity_matches_oracle():
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    frames, stds, t = orc.synthetic_stack(5, 4, 16, 12, with_std=True)
    sets = [ImageSet(value=orc.unit_from_u8(f), std=s, features=_features(ti)) for f, s, ti in zip(frames, stds, t)]
    series = ExposureSeries(input_image_sets=sets)
    series.initialize_exposure_pairs()
    icrf, _ = orc.synthetic_icrf((1.0, 1.0, 1.0))
    series.process_linearity(icrf, linearity_limit=5, use_std=True)
    ab, rel = series.collect_exposure_pair_stats()
    assert ab["means"].shape == (6, 3) and rel["stds"].shape == (6, 3)
    lo, hi = icrf[5, 0], icrf[250, 0]
    th = [orc.apply_thresholds(orc.unit_from_u8(f), s, [lo] * 3, [hi] * 3) for f, s in zip(frames, stds)]
    for s_, (rv, rs) in zip(sets, th):                                   # the image sets are left thresholded (exposure_series.py:437-441)
        np.testing.assert_array_equal(s_.measurand.val, rv)
        np.testing.assert_array_equal(s_.measurand.std, rs)
    k = 0
    with np.errstate(all="ignore"):
        for i in range(4):
            for j in range(i + 1, 4):
                if t[i] / t[j] < 0.1:
                    continue
                ad, ads, rd, rds = orc.compute_difference(th[i][0], th[i][1], th[j][0], th[j][1], t[i] / t[j])
                for res, (dv, ds) in ((ab, (ad, ads)), (rel, (rd, rds))):
                    ref = orc.dimension_statistics(dv, ds, (0, 1))
                    np.testing.assert_allclose(res["means"][k], ref["mean"], rtol=1e-11)
                    np.testing.assert_allclose(res["stds"][k], ref["std"], rtol=1e-11)
                    np.testing.assert_allclose(res["errors"][k], ref["error"], rtol=1e-11)
                k += 1
    assert k == 6


def test_host_merge_any_number_of_frames_and_tiles(heng):
    n, h, w = 40, 12, 10
    frames, stds, t = orc.synthetic_stack(3, n, h, w, with_std=True)
    t = np.asarray(t) * 2.0 ** (-(n // 2))
    icrf, diff = orc.synthetic_icrf()
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    out = heng.merge([T(f) for f in frames], t, icrf, diff, [T(s) for s in stds])
    np.testing.assert_allclose(out["val"].numpy(), ref["val"], rtol=VAL_RTOL)
    np.testing.assert_allclose(out["std"].numpy(), ref["std"], rtol=STD_RTOL)
    assert heng.plan_merge([T(f) for f in frames[:3]], t[:3], icrf).kernels == "merge_host<f64in=0,std=0,hot=0>(N=3)"
    # a row tile with the median halo gives the rows of the whole image (the sharding contract of SURVEY 8e holds for the host build too)
    rng = np.random.default_rng(0)
    dark = (rng.random((h, w, 3)) < 0.05).astype(np.uint8) * 200
    kw = dict(darks=[T(dark)] * 5, dark_min=[100] * 5, median_k=3)
    whole = heng.merge([T(f) for f in frames[:5]], t[:5], icrf, diff, [T(s) for s in stds[:5]], **kw)
    tile = heng.merge([T(f[3:10]) for f in frames[:5]], t[:5], icrf, diff, [T(s[3:10]) for s in stds[:5]], darks=[T(dark[3:10])] * 5,
                      dark_min=[100] * 5, median_k=3, height=h, row0=4, rows=5, buf_row0=3)
    assert np.array_equal(tile["val"].numpy(), whole["val"].numpy()[4:9]) and np.array_equal(tile["std"].numpy(), whole["std"].numpy()[4:9])


def test_host_backend_is_explicit_and_isolated():
    """The host library is reached only through the host classes: the HIP functional layer rejects host tensors (no CPU fallback), the
    host classes never call into libhdrmerge.so's compute entry points, and backends do not mix."""
    from camera_linearity_amd import _native as nat, engine
    from camera_linearity_amd.measurand import HipMeasurand
    icrf, _ = orc.synthetic_icrf()
    f = torch.zeros((4, 4, 3), dtype=torch.uint8)
    assert not nat.in_host_mode()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.merge([f], [1.0], icrf)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.linearize(f, None, icrf)
    before = dict(nat.hip_lib.calls)
    a = H(np.full((4, 4, 3), 0.5), np.full((4, 4, 3), 0.1))
    _ = (a + a) * 2.0
    _ = a.linearize(icrf)
    a.compute_dimension_statistics(axis=(0, 1))
    assert dict(nat.hip_lib.calls) == before                             # nothing went to the HIP library
    assert nat.host_lib().calls["hm_binary_op"] >= 2 and not nat.in_host_mode()
    with pytest.raises(TypeError, match="Invalid other type"):
        a + HipMeasurand(None, None)
    from camera_linearity_amd.image_set import ImageSet
    s = ImageSet(value=np.zeros((2, 2, 3)))
    with pytest.raises(ValueError, match="Expected type numpy"):
        s.measurand = HipMeasurand(None, None)


def test_host_library_exports_the_abi():
    """Every entry point of include/hdrmerge.h except the TIFF strip decoders (host code of libhdrmerge.so already) is exported by the host build."""
    import re
    import pathlib
    from camera_linearity_amd import _native as nat
    header = (pathlib.Path(__file__).resolve().parent.parent / "include" / "hdrmerge.h").read_text()
    declared = set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", header)) - {"hm_merge_args", "hm_tiff_lzw_decode", "hm_tiff_packbits_decode"}
    h = nat.host_lib()
    assert all(hasattr(h, name) for name in declared), [n for n in declared if not hasattr(h, n)]
    assert h.hm_version() == nat.HM_ABI_VERSION


def test_c_abi_example_host_build(tmp_path):
    """examples/merge_c_abi.c compiled with -DHM_HOST_BUILD against libhdrmerge_host.so: the same C program that drives the device
    library (tests/test_gpu_api.py::test_c_abi_example_from_plain_c), with host pointers and no HIP - and no GPU."""
    import pathlib
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    root = pathlib.Path(__file__).resolve().parent.parent
    lib = root / "camera_linearity_amd" / "lib"
    exe = tmp_path / "merge_c_abi_host"
    cmd = ["gcc", "-std=c11", "-O2", "-DHM_HOST_BUILD", str(root / "examples" / "merge_c_abi.c"), f"-I{root / 'include'}", f"-L{lib}", "-lhdrmerge_host",
           f"-Wl,-rpath,{lib}", "-lm", "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "C ABI merge OK" in r.stdout and "on host" in r.stdout


def test_fuzz_tool_self_check():
    """tools/fuzz_backends.py (device build against host build on random cases; the GPU suite runs it for real) stays runnable: host
    build against itself for three seconds."""
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "fuzz_backends.py"), "--self-check", "--seconds", "3", "--seed", "5"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout


@pytest.mark.parametrize("weighted", [False, True])
def test_host_statistics_with_infinite_values(weighted):
    """The host build on infinite values: NumPy's answers (mean +-inf / NaN by sign, nanstd NaN, the weighted formulas' inf or 0) - the same
    cases the GPU suite runs through the HIP kernels (test_statistics_with_infinite_values_answer_like_numpy)."""
    rng = np.random.default_rng(42)
    shape = (17, 13, 4)
    x = rng.normal(size=shape) + 3.0
    x[rng.random(shape) < 0.1] = np.nan
    x[3, 5, 0] = np.inf
    x[7, 1, 1] = -np.inf
    x[0, 0, 2] = np.inf; x[16, 12, 2] = -np.inf
    s = 0.01 + 0.05 * rng.random(shape) if weighted else None
    m = H(x.copy(), None if s is None else s.copy())
    with np.errstate(all="ignore"):
        for axis in ((0, 1), 0, None) + (() if weighted else (1, 2, (1, 2))):
            ref = orc.dimension_statistics(x, s, axis)
            got = m.compute_dimension_statistics(axis)
            for key in ("mean", "std"):
                g, r = np.asarray(got[key]), np.asarray(ref[key])
                fin = np.isfinite(r)
                assert np.array_equal(g[~fin], r[~fin], equal_nan=True), (axis, key, g, r)
                np.testing.assert_allclose(g[fin], r[fin], rtol=1e-11)


@pytest.mark.parametrize("use_icrf", [False, True])
def test_host_welford_matches_oracle(use_icrf):
    """video_processing.welford_algorithm(..., device="cpu"): the host build's hm_welford_update / hm_welford_finalize against the oracle's
    restatement of modules/video_processing.py:161-219 - 45 frames (two launches of 32 + 13), mean and std frames exact (uint8)."""
    from camera_linearity_amd import video_processing as vp
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (20, 30, 3)).astype(np.uint8) for _ in range(45)]
    icrf, _ = orc.synthetic_icrf()
    got = vp.welford_algorithm(iter(frames), icrf if use_icrf else None, use_std=True, device="cpu")
    ref = orc.welford(frames, icrf if use_icrf else None, True)
    assert np.array_equal(got["mean"], ref["mean"]) and np.array_equal(got["std"], ref["std"])
    only_mean = vp.welford_algorithm([iter(frames[:20]), iter(frames[20:])], None, use_std=False, device="cpu")
    assert only_mean["std"] is None and np.array_equal(only_mean["mean"], orc.welford(frames, None, False)["mean"])
    with pytest.raises(ValueError):
        vp.welford_algorithm(iter(frames[:1]), None, use_std=True, device="cpu")


@pytest.mark.parametrize("use_std", [False, True])
def test_host_energy_function_matches_oracle(use_std):
    """icrf_calibration on a host stack (initialize_channel_image_stacks(..., device="cpu")): the host build's hm_linearity_energy against
    the oracle's _energy_function / analyze_linearity (modules/ICRF_calibration_exposure.py:66-201): valid candidates to 1e-12, a
    non-monotonic and an out-of-range candidate +inf, per-pair results incl. NaN pairs."""
    from camera_linearity_amd import icrf_calibration as ic
    rng = np.random.default_rng(1)
    X, Y, N = 24, 18, 5
    dn = np.sort(rng.integers(0, 256, (X, Y, N)).astype(np.uint8), axis=2)
    sd = 0.004 * (1 + rng.random((X, Y, N))) if use_std else None
    t = 1e-3 * 2.0 ** np.arange(N)
    cands = np.stack([np.linspace(0, 1, 256) ** g for g in (0.8, 1.0, 2.2, 1.7)])
    bad = cands[1].copy(); bad[100] = bad[99]                        # not strictly increasing
    cands = np.vstack([cands, bad[None]])
    valid = np.array([orc.candidate_valid(c) for c in cands])
    assert list(valid) == [True, True, True, True, False]
    vt, st = torch.from_numpy(dn), None if sd is None else torch.from_numpy(sd)
    e, pairs = ic._engine_for(vt).linearity_energy(vt, st, t, cands, 5, 250, valid, True, return_pairs=True)
    ref = np.array([orc.energy_function(c, dn, sd, 5, 250, t) for c in cands])
    assert np.isinf(e.numpy()[4]) and np.isinf(ref[4])
    np.testing.assert_allclose(e.numpy()[:4], ref[:4], rtol=1e-12)
    ref_pairs = orc.analyze_linearity_pairs(cands[2][dn], sd, cands[2][5], cands[2][250], True, t)
    np.testing.assert_allclose(pairs.numpy()[2], ref_pairs, rtol=1e-12, equal_nan=True)
    one = ic.analyze_linearity(vt, st, cands[2], 5, 250, True, t)
    np.testing.assert_allclose(one.numpy(), ref_pairs, rtol=1e-12, equal_nan=True)


def test_host_calibration_recovers_response():
    """icrf_calibration end to end on the HOST build (initialize_channel_image_stacks(..., device="cpu")): SciPy's differential evolution
    over PCA coefficients, the population's energies from one hm_linearity_energy call per generation, recovers a synthetic camera
    response (energy drops by more than 10x from the start point) - the GPU suite's test_calibration_recovers_response without a GPU."""
    from camera_linearity_amd import icrf_calibration as cal
    rng = np.random.default_rng(21)
    X, Y, N = 24, 24, 5
    t = 1e-3 * 2.0 ** np.arange(N)
    xs = np.linspace(0, 1, 256)
    pca = np.stack([np.sin(np.pi * (m + 1) * xs) / (m + 1) for m in range(3)], axis=1) * 0.1
    mean_icrf = xs ** 2.0
    true_params = np.array([0.6, -0.3, 0.2])
    true_icrf, ok = cal.candidate_icrfs(true_params, mean_icrf, pca)
    assert ok[0]
    rad = rng.random((X, Y)) * 2.5 / t[-1]
    lin = np.clip(rad[..., None] * t, 0, 1)
    dn = np.clip(np.around(np.interp(lin, true_icrf[0], xs) * 255), 0, 255).astype(np.uint8)
    stacks, stds, tt = cal.initialize_channel_image_stacks([dn[:, :, None, i].repeat(1, 2) for i in range(N)], t, None, 1, device="cpu")
    assert not stacks[0].is_cuda and stacks[0].shape == (X, Y, N)
    e0 = cal._energy_function(np.zeros(3), mean_icrf, pca, stacks[0], None, 5, 250, True, tt)
    e_true = cal._energy_function(true_params, mean_icrf, pca, stacks[0], None, 5, 250, True, tt)
    assert e_true < e0 / 10
    icrf, e, n_it = cal.solve_channel(mean_icrf, pca, stacks[0], None, tt, -1.0, 1.0, seed=7, max_iterations=30, vectorized=True)
    assert icrf.shape == (256,) and e < e0 / 10, (e, e0, e_true)


def test_flat_roi_means_are_computed_once_per_flat(golden):
    """One flat field serves many exposure series: its ROI means (a reduction + a copy to the host) are remembered on the flat's ImageSet and
    computed again only when the flat image or the ROI changes."""
    from camera_linearity_amd import _native as nat, settings as gs
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    g = golden("merge_full")
    old = (gs.DARK_THRESHOLD, gs.FF_MID_PERCENTAGE, gs.MEDIAN_FILTER_KERNEL_SIZE)
    gs.configure(DARK_THRESHOLD=float(g["dark_threshold"]), FF_MID_PERCENTAGE=float(g["ff_mid"]), MEDIAN_FILTER_KERNEL_SIZE=int(g["median_k"]))
    try:
        flat = ImageSet(value=g["flat"], std=g["flat_std"], features=dict(_features(0.01), subject="flat"))
        calls = nat.host_lib().calls

        def run():
            sets = [ImageSet(value=g["frames"][i], std=g["stds"][i], features=_features(t)) for i, t in enumerate(g["exposures"])]
            series = ExposureSeries(input_image_sets=sets)
            series.process_HDR_image(g["icrf"], g["icrf_diff"], flat_list=[flat])
            return series.merged_image_set.host_arrays()
        before = calls["hm_roi_mean_u8"] + calls["hm_roi_mean_f64"]
        v1, s1 = run()
        mid = calls["hm_roi_mean_u8"] + calls["hm_roi_mean_f64"]
        v2, s2 = run()
        after = calls["hm_roi_mean_u8"] + calls["hm_roi_mean_f64"]
        assert mid - before == 2 and after == mid                      # value and std means once; the second series reuses them
        assert np.array_equal(v1, v2) and np.array_equal(s1, s2)
        gs.configure(FF_MID_PERCENTAGE=0.5)                            # another ROI: computed again
        run()
        assert calls["hm_roi_mean_u8"] + calls["hm_roi_mean_f64"] == after + 2
    finally:
        gs.configure(DARK_THRESHOLD=old[0], FF_MID_PERCENTAGE=old[1], MEDIAN_FILTER_KERNEL_SIZE=old[2])
