"""GPU tests of the rows SURVEY.md 8(f) ranks after the merge: the Welford mean / std producer
(modules/video_processing.py:161-219) and the ICRF-calibration energy function
(modules/ICRF_calibration_exposure.py:66-201). Both go through the C ABI (hm_welford_*, hm_linearity_energy) and
are checked against the oracle and the fixtures the reference itself produced."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hdr_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from camera_linearity_amd import engine
    return engine


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------------------------------ Welford
def test_welford_reference_fixture(eng, golden):
    from camera_linearity_amd.video_processing import welford_algorithm
    g = golden("welford")
    r = welford_algorithm(g["clip"], None, True)
    assert r["mean"].dtype == np.uint8 and np.array_equal(r["mean"], g["mean"])
    assert np.array_equal(r["std"], g["std"])
    r = welford_algorithm(list(g["clip"]) + [None], None, False, frames_per_launch=5)
    assert r["std"] is None and np.array_equal(r["mean"], g["mean"])


@pytest.mark.parametrize("shape,n,with_icrf,batch", [((12, 20, 3), 37, False, 16), ((7, 9, 3), 45, True, 32),
                                                      ((5, 5, 1), 9, False, 4), ((3, 11, 4), 33, True, 7),
                                                      ((1, 1, 3), 2, False, 1), ((16, 64, 3), 700, False, 32), ((16, 64, 3), 301, True, 32)])
def test_welford_state_bit_exact(eng, shape, n, with_icrf, batch):
    """The running float64 mean and M2 carry the oracle's bits (same op order, IEEE division), for every batch split."""
    rng = np.random.default_rng(sum(shape) + n)
    clip = rng.integers(0, 256, (n,) + shape, dtype=np.uint8)
    icrf = np.linspace(0, 1, 256)[:, None] ** np.linspace(1.6, 2.4, shape[-1])[None, :] if with_icrf else None
    mean_ref, m2_ref, cnt = orc.welford_state(list(clip), icrf, True)
    mean = torch.zeros(shape, dtype=torch.float64, device="cuda")
    m2 = torch.zeros(shape, dtype=torch.float64, device="cuda")
    count = 0
    frames = [dev(f) for f in clip]
    for k0 in range(0, n, batch):
        count = eng.welford_update(frames[k0:k0 + batch], count, mean, m2, icrf)
    assert count == cnt
    assert np.array_equal(mean.cpu().numpy(), mean_ref)
    assert np.array_equal(m2.cpu().numpy(), m2_ref)
    mu8, sd8 = eng.welford_finalize(mean, m2, count)
    ref_mu8, ref_sd8 = orc.welford_finalize(mean_ref, m2_ref, cnt)
    assert np.array_equal(mu8.cpu().numpy(), ref_mu8) and np.array_equal(sd8.cpu().numpy(), ref_sd8)
    # mean only
    mean2 = torch.zeros(shape, dtype=torch.float64, device="cuda")
    eng.welford_update(frames, 0, mean2, None, icrf)
    assert np.array_equal(mean2.cpu().numpy(), mean_ref)


@pytest.mark.parametrize("case", ["signed_table", "tiny_entry", "tiny_state", "huge_state", "nan_state", "zero_runs"])
def test_welford_fast_division_range(eng, case):
    """The fast delta / n (reciprocal from the host + two FMAs) is taken only where its range argument holds - table entries 0 or in
    [2^-500, 2^500], incoming mean 0 or in [2^-540, 2^500] - and the full division otherwise: both give the oracle's bits. Signed tables
    make the running mean cancel (the worst case of the argument), runs of zero-valued frames shrink it."""
    rng = np.random.default_rng(len(case))
    shape, n = (9, 14, 3), 41
    clip = rng.integers(0, 256, (n,) + shape, dtype=np.uint8)
    icrf = np.linspace(0, 1, 256)[:, None] ** np.array([1.7, 2.0, 2.3])[None, :]
    mean0 = np.zeros(shape)
    if case == "signed_table":
        icrf = icrf - 0.37                                                 # both signs: m (1 - 1/n) + f/n cancels
        icrf[100] = -icrf[101]
    elif case == "tiny_entry":
        icrf[7, 1] = 1e-200                                                # outside [2^-500, 2^500]: the whole launch takes the exact path
    elif case == "tiny_state":
        mean0[2, 3, 1] = 1e-250
        mean0[4, 5, 0] = -3e-300
    elif case == "huge_state":
        mean0[1, 1, 2] = 1e200
    elif case == "nan_state":
        mean0[0, 0, 0] = np.nan
    elif case == "zero_runs":
        clip[5:30] = 0                                                     # ICRF[0] = 0: twenty-five frames of f = 0
        icrf[0] = 0.0
    count0 = 3 if case.endswith("state") else 0
    m2_0 = np.abs(mean0) * 0.0
    with np.errstate(all="ignore"):
        mean_ref, m2_ref, cnt = orc.welford_state(list(clip), icrf, True, mean=mean0.copy(), m2=m2_0.copy(), count=count0)
    mean, m2 = dev(mean0), dev(m2_0)
    frames = [dev(f) for f in clip]
    count = count0
    for k0 in range(0, n, 16):
        count = eng.welford_update(frames[k0:k0 + 16], count, mean, m2, icrf)
    assert count == cnt
    np.testing.assert_array_equal(mean.cpu().numpy(), mean_ref)
    np.testing.assert_array_equal(m2.cpu().numpy(), m2_ref)


def test_welford_unaligned_and_odd(eng):
    """Odd element count and frames at odd byte offsets take the scalar path: same bits."""
    rng = np.random.default_rng(5)
    n, shape = 6, (3, 5, 3)                       # 45 elements (odd)
    buf = torch.as_tensor(rng.integers(0, 256, n * 45 + 1, dtype=np.uint8), device="cuda")
    frames = [buf[1 + k * 45: 1 + (k + 1) * 45].view(shape) for k in range(n)]       # odd offsets
    clip = [f.cpu().numpy() for f in frames]
    mean_ref, m2_ref, _ = orc.welford_state(clip, None, True)
    mean = torch.zeros(shape, dtype=torch.float64, device="cuda")
    m2 = torch.zeros(shape, dtype=torch.float64, device="cuda")
    eng.welford_update(frames, 0, mean, m2)
    assert np.array_equal(mean.cpu().numpy(), mean_ref) and np.array_equal(m2.cpu().numpy(), m2_ref)


def test_welford_errors(eng):
    from camera_linearity_amd.video_processing import welford_algorithm
    with pytest.raises(ValueError):
        welford_algorithm([], None, False)
    with pytest.raises(ValueError):
        welford_algorithm([np.zeros((2, 2, 3), np.uint8)], None, True)           # std needs two frames
    with pytest.raises(TypeError):
        welford_algorithm([np.zeros((2, 2, 3), np.float64)], None, False)
    with pytest.raises(ValueError):
        welford_algorithm([np.zeros((2, 2, 3), np.uint8), np.zeros((2, 3, 3), np.uint8)], None, False)


def test_welford_large_constant_and_linearity(eng):
    """Size-independent properties at a full-frame size: a constant clip has mean == the constant and M2 == 0; in
    either frame order the running mean / M2 equal the two-pass values formed from the exact integer sum."""
    H, W = 1024, 1536
    const = torch.full((H, W, 3), 137, dtype=torch.uint8, device="cuda")
    mean = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    m2 = torch.zeros_like(mean)
    c = eng.welford_update([const] * 20, 0, mean, m2)
    assert c == 20 and bool((mean == 137 / 255).all()) and bool((m2 == 0).all())
    g = torch.Generator(device="cuda").manual_seed(3)
    clip = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(12)]
    exact = torch.stack(clip).to(torch.int64).sum(dim=0).double() / (12 * 255)         # exact integer sum
    sq = (torch.stack(clip).double() / 255 - exact).pow(2).sum(dim=0)
    for order in (clip, clip[::-1]):
        mean.zero_(); m2.zero_()
        c = eng.welford_update(order, 0, mean, m2)
        assert float((mean - exact).abs().max()) < 1e-15
        assert float((m2 - sq).abs().max()) < 1e-13


# ------------------------------------------------------------------------------------------------ energy function
def test_energy_reference_fixture(eng, golden):
    from camera_linearity_amd import icrf_calibration as cal
    g = golden("energy")
    lo, up = int(g["lower"]), int(g["upper"])
    dn, sd = dev(g["dn"]), dev(g["sd"])
    for std, key, pkey in ((None, "energy_plain", "pairs_plain"), (sd, "energy_std", "pairs_std")):
        # whole population in one launch
        e = cal.energy_function_batch(g["params"], g["mean_icrf"], g["pca"], dn, std, lo, up, True, g["exposures"])
        np.testing.assert_allclose(e, g[key], rtol=1e-12)
        assert np.array_equal(np.isinf(e), np.isinf(g[key]))
        # SciPy's vectorized calling convention (n_params, S)
        e2 = cal._energy_function(g["params"].T, g["mean_icrf"], g["pca"], dn, std, lo, up, True, g["exposures"])
        assert np.array_equal(e2, e)
        # one candidate at a time: the reference's signature
        for b in (0, 1, 3):
            eb = cal._energy_function(g["params"][b], g["mean_icrf"], g["pca"], dn, std, lo, up, True, g["exposures"])
            assert isinstance(eb, float) and eb == e[b]
        icrfs, valid = cal.candidate_icrfs(g["params"], g["mean_icrf"], g["pca"])
        assert np.array_equal(icrfs, g["icrfs"])
        _, pairs = eng.linearity_energy(dn, std, g["exposures"], icrfs, lo, up, valid, True, return_pairs=True)
        pairs = pairs.cpu().numpy()
        np.testing.assert_allclose(pairs[valid], g[pkey][valid], rtol=1e-12, equal_nan=True)
        assert np.isnan(pairs[~valid]).all()
    # absolute mode of analyze_linearity
    c = g["icrfs"][0]
    np.testing.assert_allclose(cal.analyze_linearity(dn, None, c, lo, up, False, g["exposures"]).cpu().numpy(), g["abs_plain"],
                               rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(cal.analyze_linearity(dn, sd, c, lo, up, False, g["exposures"]).cpu().numpy(), g["abs_std"],
                               rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize("X,Y,N,with_std", [(27, 27, 7, False), (27, 27, 7, True), (65, 129, 3, True), (9, 5, 12, False),
                                            (300, 200, 4, True), (1, 1, 2, False)])
def test_energy_vs_oracle(eng, X, Y, N, with_std):
    rng = np.random.default_rng(X * 7 + N)
    dn = rng.integers(0, 256, (X, Y, N), dtype=np.uint8)
    sd = 0.004 * (1 + rng.random((X, Y, N))) if with_std else None
    t = 1e-3 * 1.7 ** np.arange(N)
    gammas = np.array([1.0, 1.5, 2.2, 3.0])
    icrfs = np.linspace(0, 1, 256)[None, :] ** gammas[:, None]
    icrfs[:, -1] = 1.0
    e, pairs = eng.linearity_energy(dev(dn), None if sd is None else dev(sd), t, icrfs, 5, 250, None, True, return_pairs=True)
    for b in range(len(gammas)):
        ref_pairs = orc.analyze_linearity_pairs(icrfs[b][dn], sd, icrfs[b][5], icrfs[b][250], True, t)
        np.testing.assert_allclose(pairs[b].cpu().numpy(), ref_pairs, rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(float(e[b]), orc.energy_function(icrfs[b], dn, sd, 5, 250, t), rtol=1e-12)


def test_energy_all_masked_is_inf(eng):
    dn = np.zeros((4, 4, 3), dtype=np.uint8)                        # every DN below `lower`
    icrf = np.linspace(0, 1, 256)
    e = eng.linearity_energy(dev(dn), None, [1.0, 2.0, 4.0], icrf[None], 5, 250)
    assert np.isinf(float(e[0]))
    assert np.isinf(orc.energy_function(icrf, dn, None, 5, 250, [1.0, 2.0, 4.0]))


def test_energy_exact_linear_response_is_zero(eng):
    """Size-independent property at a large size: a perfectly linear camera seen through the identity ICRF has zero
    energy for the pairs whose samples stay inside the limits (here: all of them, by construction)."""
    X, Y = 1024, 1024
    base = torch.randint(10, 31, (X, Y), dtype=torch.uint8, device="cuda")
    dn = torch.stack([base, base * 2, base * 4, base * 8], dim=-1).contiguous()       # DN proportional to exposure
    icrf = np.arange(256) / 255.0
    icrf[-1] = 1.0
    e, pairs = eng.linearity_energy(dn, None, [1.0, 2.0, 4.0, 8.0], icrf[None], 5, 250, None, True, return_pairs=True)
    assert float(pairs.abs().max()) < 1e-15 and float(e[0]) < 1e-15


def test_energy_errors(eng):
    dn = torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda")
    icrf = np.linspace(0, 1, 256)[None]
    with pytest.raises(ValueError):
        eng.linearity_energy(dn[0], None, [1.0, 2.0, 4.0], icrf, 5, 250)            # not 3-D
    with pytest.raises(ValueError):
        eng.linearity_energy(dn, None, [1.0, 2.0], icrf, 5, 250)                    # exposures do not match N
    with pytest.raises(TypeError):
        eng.linearity_energy(dn.double(), None, [1.0, 2.0, 4.0], icrf, 5, 250)
    with pytest.raises(ValueError):
        eng.linearity_energy(dn[:, :, :1].contiguous(), None, [1.0], icrf, 5, 250)  # a single frame has no pairs


def test_calibration_recovers_response(eng):
    """End to end: differential evolution over PCA coefficients with the population evaluated per launch recovers a
    synthetic camera response (energy drops by > 10x from the start point), in both solver modes."""
    from camera_linearity_amd import icrf_calibration as cal
    rng = np.random.default_rng(21)
    X, Y, N = 40, 40, 6
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
    stacks, stds, tt = cal.initialize_channel_image_stacks([dn[:, :, None, i].repeat(1, 2) for i in range(N)], t, None, 1)
    assert stacks[0].shape == (X, Y, N) and np.array_equal(stacks[0].cpu().numpy(), dn)
    e0 = cal._energy_function(np.zeros(3), mean_icrf, pca, stacks[0], None, 5, 250, True, tt)
    e_true = cal._energy_function(true_params, mean_icrf, pca, stacks[0], None, 5, 250, True, tt)
    assert e_true < e0 / 10
    for vectorized, iters in ((True, 40), (False, 6)):
        icrf, e, n_it = cal.solve_channel(mean_icrf, pca, stacks[0], None, tt, -1.0, 1.0, seed=7, max_iterations=iters,
                                          vectorized=vectorized)
        assert icrf.shape == (256,) and n_it <= iters
        assert e < e0 / (10 if vectorized else 2), (vectorized, e, e0, e_true)


# ------------------------------------------------------------------------------------------------ on-disk workflow (8f-4)
def test_from_dir_path_tiff_workflow(eng, tmp_path):
    """The reference's file workflow without cv2: an exposure series of 8-bit BGR TIFFs with float64 ' STD.tif' companions
    (file-name grammar of modules/image_set.py:542-568) -> ExposureSeries.from_dir_path -> process_HDR_image ->
    save_64bit(is_HDR=True) -> files that read back to the merged arrays; the merge equals the oracle on the same arrays."""
    from camera_linearity_amd import tiff_io
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    frames, stds, t = orc.synthetic_stack(9, 4, 40, 56, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    for f, s, ti in zip(frames, stds, t):
        name = f"{ti * 1000:g}ms bf 5x sample.tif"
        tiff_io.imwrite(tmp_path / name, f)
        tiff_io.imwrite(tmp_path / name.replace(".tif", " STD.tif"), s)
    sets = ImageSet.multiple_from_path(tmp_path, use_cupy=True)
    assert len(sets) == 4 and sorted(s.features["exposure"] for s in sets) == sorted(t.tolist())
    assert all(s.features["illumination"] == "bf" and s.features["magnification"] == "5x" for s in sets)
    series_list = ExposureSeries.from_dir_path(tmp_path, use_cupy=True)
    series = series_list[0] if isinstance(series_list, list) else series_list
    assert len(series.input_image_sets) == 4
    series.process_HDR_image(icrf, diff)
    val, std = series.merged_image_set.host_arrays()
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    np.testing.assert_allclose(val, ref["val"], rtol=1e-12)
    np.testing.assert_allclose(std, ref["std"], rtol=1e-9)
    out = tmp_path / "out" / "bf 5x sample.tif"
    series.merged_image_set.save_64bit(out, is_HDR=True)
    back = tiff_io.imread(tmp_path / "out" / "bf 5x sample HDR.tif", tiff_io.IMREAD_UNCHANGED)
    back_std = tiff_io.imread(tmp_path / "out" / "bf 5x sample HDR STD.tif", tiff_io.IMREAD_UNCHANGED)
    assert back.dtype == np.float64 and np.array_equal(back, val) and np.array_equal(back_std, std)
    series.merged_image_set.save_8bit(tmp_path / "out8" / "bf 5x sample.tif")
    v8 = tiff_io.imread(tmp_path / "out8" / "bf 5x sample.tif")
    assert v8.dtype == np.uint8 and np.array_equal(v8, np.around(val / max(val.max(), 1.0) * 255).astype(np.uint8))
    # the Welford producer's output files (video_processing.py:222-236)
    from camera_linearity_amd.video_processing import process_video
    clip = np.random.default_rng(1).integers(0, 256, (6, 10, 12, 3), dtype=np.uint8)
    ret = process_video(clip, tmp_path / "clip 10ms.avi", None, True)
    assert np.array_equal(tiff_io.imread(tmp_path / "clip 10ms.mean.tif"), ret["mean"])
    assert np.array_equal(tiff_io.imread(tmp_path / "clip 10ms.std.tif"), ret["std"])


def _write_series(tiff_io, folder, frames, stds, t, tag="bf 5x sample"):
    folder.mkdir(parents=True, exist_ok=True)
    for f, s, ti in zip(frames, stds, t):
        name = f"{ti * 1000:g}ms {tag}.tif"
        tiff_io.imwrite(folder / name, f)
        if s is not None:
            tiff_io.imwrite(folder / name.replace(".tif", " STD.tif"), s)


def test_std_images_load_after_value_images(eng, tmp_path):
    """The reference workflow order: series.load_value_images() first, then process_HDR_image. The merge loop of the
    reference calls load_std_image() for every frame whatever is in memory (modules/exposure_series.py:376-377), so the
    merged uncertainty must still be there - and a series where only some frames have a ' STD.tif' is an error, not a
    silent value-only merge."""
    from camera_linearity_amd import tiff_io
    from camera_linearity_amd.exposure_series import ExposureSeries
    frames, stds, t = orc.synthetic_stack(12, 3, 24, 40, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    _write_series(tiff_io, tmp_path / "a", frames, stds, t)
    series = ExposureSeries.from_dir_path(tmp_path / "a", use_cupy=True)[0]
    series.load_value_images()
    assert all(s.measurand.std is None for s in series.input_image_sets)
    series.process_HDR_image(icrf, diff)
    val, std = series.merged_image_set.host_arrays()
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    assert std is not None
    np.testing.assert_allclose(val, ref["val"], rtol=1e-12)
    np.testing.assert_allclose(std, ref["std"], rtol=1e-9)
    _write_series(tiff_io, tmp_path / "b", frames, [stds[0], None, stds[2]], t)
    partial = ExposureSeries.from_dir_path(tmp_path / "b", use_cupy=True)[0]
    with pytest.raises(ValueError):
        partial.process_HDR_image(icrf, diff)
    partial.process_HDR_image(icrf, diff, use_std=False)
    np.testing.assert_allclose(partial.merged_image_set.host_arrays()[0], ref["val"], rtol=1e-12)


def test_process_hdr_image_default_arguments(eng, tmp_path):
    """process_HDR_image() with no arguments, as a caller of the reference writes it (modules/exposure_series.py:399-409):
    ICRF from settings.ICRF_CALIBRATED_FILE, dark frames from settings.DEFAULT_DARK_PATH, flat fields from
    settings.DEFAULT_FLAT_PATH; equal to the explicit-argument call and to the oracle. Without a configured ICRF file it raises."""
    from camera_linearity_amd import settings, tiff_io
    from camera_linearity_amd.exposure_series import ExposureSeries
    from camera_linearity_amd.image_set import ImageSet
    n, h, w = 3, 32, 48
    rng = np.random.default_rng(31)
    t = np.array([0.02, 0.1, 0.4])                                        # two frames at or above DARK_THRESHOLD = 0.05 s
    rad = rng.random((h, w, 3)) * 4
    frames = [np.clip(np.around(rad * ti * 255 / (4 * t[1])), 0, 255).astype(np.uint8) for ti in t]
    stds = [0.004 * (1 + rng.random((h, w, 3))) for _ in t]
    icrf, diff = orc.synthetic_icrf()
    dark = rng.integers(0, 4, size=(h, w, 3)).astype(np.uint8)
    dark[rng.random((h, w, 3)) < 0.02] = 200
    flat = np.clip(np.around(255 * (0.8 + 0.05 * rng.random((h, w, 3)))), 0, 255).astype(np.uint8)
    flat_std = np.full((h, w, 3), 0.002)
    _write_series(tiff_io, tmp_path / "series", frames, stds, t)
    _write_series(tiff_io, tmp_path / "darks", [dark, dark, dark], [None] * 3, [0.05, 0.1, 0.4], tag="bf 5x dark")
    _write_series(tiff_io, tmp_path / "flats", [flat], [flat_std], [0.1], tag="bf 5x flat")
    np.savetxt(tmp_path / "icrf.txt", icrf)
    series = ExposureSeries.from_dir_path(tmp_path / "series", use_cupy=True)[0]
    with pytest.raises(ValueError):
        series.process_HDR_image()
    saved = {k: getattr(settings, k) for k in ("ICRF_CALIBRATED_FILE", "DEFAULT_DARK_PATH", "DEFAULT_FLAT_PATH")}
    try:
        settings.configure(ICRF_CALIBRATED_FILE=tmp_path / "icrf.txt", DEFAULT_DARK_PATH=tmp_path / "darks", DEFAULT_FLAT_PATH=tmp_path / "flats")
        series.process_HDR_image()
        val, std = series.merged_image_set.host_arrays()
    finally:
        settings.configure(**saved)
    explicit = ExposureSeries.from_dir_path(tmp_path / "series", use_cupy=True)[0]
    explicit.process_HDR_image(np.loadtxt(tmp_path / "icrf.txt"), None, dark_list=ImageSet.multiple_from_path(tmp_path / "darks", use_cupy=True),
                               flat_list=ImageSet.multiple_from_path(tmp_path / "flats", use_cupy=True))
    ev, es = explicit.merged_image_set.host_arrays()
    assert np.array_equal(val, ev) and np.array_equal(std, es)
    fval = orc.unit_from_u8(flat)
    dv = orc.unit_from_u8(dark)
    ref = orc.merge(frames, t, np.loadtxt(tmp_path / "icrf.txt"), orc.icrf_derivative(np.loadtxt(tmp_path / "icrf.txt")), stds=stds,
                    darks=[None, dv, dv], dark_threshold=settings.DARK_THRESHOLD, median_k=3, flat=fval, flat_std=flat_std,
                    ff_mean=orc.flat_roi_mean(fval, h, w, 0.2), ff_std_mean=orc.flat_roi_mean(flat_std, h, w, 0.2))
    np.testing.assert_allclose(val, ref["val_ff"], rtol=1e-12)
    np.testing.assert_allclose(std, ref["std_ff"], rtol=1e-9)


# ------------------------------------------------------------------------------------------------ host-to-host pipeline
@pytest.mark.parametrize("with_std,depth,count", [(False, 2, 5), (True, 3, 4), (False, 2, 1)])
def test_merge_pipeline_matches_oracle(eng, with_std, depth, count):
    """MergePipeline (H2D, merge and D2H of consecutive stacks overlapped on three streams): every stack's result equals the
    oracle's, in order, for more stacks than slots (slot reuse) and for a single stack."""
    from camera_linearity_amd.pipeline import MergePipeline
    n, h, w = 4, 40, 64
    icrf, diff = orc.synthetic_icrf()
    stacks = [orc.synthetic_stack(40 + k, n, h, w, with_std=with_std) for k in range(count)]
    t = stacks[0][2]
    pipe = MergePipeline(n, h, w, t, icrf, diff, with_std=with_std, depth=depth)
    res = pipe.merge_many([s[0] for s in stacks], [s[1] for s in stacks] if with_std else None)
    assert len(res) == count
    for (val, std), (frames, stds, _) in zip(res, stacks):
        ref = orc.merge(frames, t, icrf, diff, stds=stds)
        np.testing.assert_allclose(val, ref["val"], rtol=1e-12)
        if with_std:
            np.testing.assert_allclose(std, ref["std"], rtol=1e-9)
        else:
            assert std is None
    # the generator interface: results arrive in order and the pipeline can be driven again
    def fill(k, fv, sv):
        if k >= 3:
            return False
        for d, src in zip(fv, stacks[0][0]):
            np.copyto(d, src)
        if sv is not None:
            for d, src in zip(sv, stacks[0][1]):
                np.copyto(d, src)
        return True
    seen = [k for k, _, _ in pipe.run(fill)]
    assert seen == [0, 1, 2]
    with pytest.raises(ValueError):
        MergePipeline(n, h, w, t, icrf, diff, depth=1)


def test_calibrate_flats_and_darks_from_disk(eng, tmp_path):
    """calibrate_flats / calibrate_dark_frames (image_set.py:504-539): bias subtraction through Measurand.__sub__ and 8-bit
    saving, directories as arguments."""
    from camera_linearity_amd import tiff_io
    from camera_linearity_amd.image_set import calibrate_dark_frames, calibrate_flats
    rng = np.random.default_rng(3)
    raw_dark, dark, raw_flat, flat = (tmp_path / d for d in ("raw_dark", "dark", "raw_flat", "flat"))
    for d in (raw_dark, raw_flat):
        d.mkdir()
    bias = rng.integers(0, 8, (20, 30, 3), dtype=np.uint8)
    darks = {"1ms": bias, "50ms": (bias + rng.integers(0, 20, (20, 30, 3))).astype(np.uint8)}
    for name, img in darks.items():
        tiff_io.imwrite(raw_dark / f"{name} dark.tif", img)
    flat_img = rng.integers(150, 240, (20, 30, 3), dtype=np.uint8)
    tiff_io.imwrite(raw_flat / "10ms flat.tif", flat_img)
    out = calibrate_dark_frames(raw_dark, dark)
    assert len(out) == 2
    got = tiff_io.imread(dark / "50ms dark.tif")
    want = np.around((darks["50ms"].astype(np.float64) / 255 - bias.astype(np.float64) / 255) * 255).astype(np.uint8)
    assert np.array_equal(got, want)
    assert not tiff_io.imread(dark / "1ms dark.tif").any()                       # bias minus itself
    calibrate_flats(raw_dark, raw_flat, flat)
    got = tiff_io.imread(flat / "10ms flat.tif")
    want = np.around((flat_img.astype(np.float64) / 255 - bias.astype(np.float64) / 255) * 255).astype(np.uint8)
    assert np.array_equal(got, want)
