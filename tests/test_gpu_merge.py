"""GPU parity tests of the fused merge and its per-frame kernels, through the C ABI (libhdrmerge.so).

Checker = oracle/hdr_oracle.py (NumPy restatement pinned against the reference, tests/test_oracle_golden.py)
and the golden vectors produced by running the reference (tests/golden/*.npz).

Tolerances (north_star: LUT index bit-exact, float64 radiance/uncertainty within 1e-6 relative):
  index ............. exact
  val (uint8 frames)  rtol 1e-12   (kernel sums w*g/t then divides by S; the reference divides per frame)
  std ............... rtol 1e-9    (the A-term of exposure_series.py:389 cancels; 1/S multiplies replace divides)
  float64 frames .... rtol 1e-11   (device exp() vs NumPy `np.e ** x`, <= few ulp)
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hdr_oracle as orc  # noqa: E402

VAL_RTOL, STD_RTOL, F64_RTOL = 1e-12, 1e-9, 1e-11


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from camera_linearity_amd import engine
    return engine


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), device="cuda")


def host(t):
    return t.cpu().numpy()


def close(a, b, rtol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=0)


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("name", ["merge_identity", "merge_std", "merge_ramp"])
def test_merge_golden_u8(eng, golden, name):
    g = golden(name)
    frames = [dev(f) for f in g["frames"]]
    stds = [dev(s) for s in g["stds"]] if "stds" in g else None
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds, want_sum_w=True)
    close(host(out["sum_w"]), g["S"], 1e-15 * 8)
    close(host(out["val"]), g["val"], VAL_RTOL)
    if stds is not None:
        close(host(out["std"]), g["std"], STD_RTOL)


def test_merge_golden_float_frames(eng, golden):
    g = golden("merge_float")
    frames = [dev(f) for f in g["frames_f64"]]
    stds = [dev(s) for s in g["stds"]]
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds)
    close(host(out["val"]), g["val"], F64_RTOL)
    close(host(out["std"]), g["std"], STD_RTOL)


def _dark_setup(eng, g, dark_arrays):
    darks, mins = [], []
    for t in g["exposures"]:
        j, sc = orc.select_dark(float(t), [float(x) for x in g["dark_exposures"]], float(g["dark_threshold"]))
        darks.append(None if j < 0 else dev(dark_arrays[j]))
        mins.append(256 if j < 0 else eng.dark_min_dn(sc, float(g["dark_threshold"])))
    return darks, mins


def test_merge_golden_full_corrections(eng, golden):
    g = golden("merge_full")
    frames = [dev(f) for f in g["frames"]]
    stds = [dev(s) for s in g["stds"]]
    darks, mins = _dark_setup(eng, g, [g["dark16"], g["dark32"], g["dark64"]])
    flat, flat_std = dev(g["flat"]), dev(g["flat_std"])
    h, w = g["flat"].shape[:2]
    x0, x1, y0, y1 = eng.flat_roi_bounds(h, w, float(g["ff_mid"]))
    m = host(eng.roi_mean(flat, x0, x1, y0, y1))
    s = host(eng.roi_mean(flat_std, x0, x1, y0, y1))
    close(m, g["ff_mean"], 1e-14)
    close(s, g["ff_std_mean"], 1e-14)
    # hot-pixel filter only
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds, darks=darks, dark_min=mins,
                    median_k=int(g["median_k"]))
    close(host(out["val"]), g["val"], VAL_RTOL)
    close(host(out["std"]), g["std"], STD_RTOL)
    # + flat field
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds, darks=darks, dark_min=mins,
                    median_k=int(g["median_k"]), flat=flat, flat_std=flat_std, ff_mean=m, ff_std_mean=s)
    close(host(out["val"]), g["val_ff"], VAL_RTOL)
    close(host(out["std"]), g["std_ff"], STD_RTOL)
    # no corrections
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds)
    close(host(out["val"]), g["val_nohot"], VAL_RTOL)
    close(host(out["std"]), g["std_nohot"], STD_RTOL)
    # standalone normalize_by_map on the reference's merged image
    nv, ns = eng.normalize_by_map(dev(g["val"]), dev(g["std"]), flat, flat_std, m, s)
    close(host(nv), g["val_ff"], 1e-14)
    close(host(ns), g["std_ff"], 1e-13)


def test_merge_golden_scaled_darks(eng, golden):
    g = golden("merge_dark_scaled")
    frames = [dev(f) for f in g["frames"]]
    stds = [dev(s) for s in g["stds"]]
    darks, mins = _dark_setup(eng, g, list(g["darks"]))
    out = eng.merge(frames, g["exposures"], g["icrf"], g["icrf_diff"], stds, darks=darks, dark_min=mins,
                    median_k=int(g["median_k"]))
    close(host(out["val"]), g["val"], VAL_RTOL)
    close(host(out["std"]), g["std"], STD_RTOL)


# ------------------------------------------------------------------ per-frame kernels
def test_linearize_index_bit_exact_and_values(eng, golden):
    g = golden("merge_float")
    v = g["frames_f64"][0]
    val, std, idx = eng.linearize(dev(v), dev(g["stds"][0]), g["icrf"], g["icrf_diff"], return_index=True)
    assert np.array_equal(host(idx), g["idx"][0])             # bit-exact uint8 LUT index incl. .5 ties and wraps
    assert np.array_equal(host(val), g["lin0_val"])           # a gather: exact
    close(host(std), g["lin0_std"], 1e-15)
    for i in (1, 2):
        _, _, idx = eng.linearize(dev(g["frames_f64"][i]), None, g["icrf"], return_index=True)
        assert np.array_equal(host(idx), g["idx"][i])


def test_negative_float_inputs_wrap_like_numpy(eng, golden):
    """Bias-subtracted frames can be negative (modules/image_set.py:520,538): around(v * 255).astype(uint8) wraps them to 253..255
    (modules/measurand.py:503,531). The reference-generated fixture pins -0.25/255 .. -3/255 incl. the -0.5 / -1.5 / -2.5 ties, -1e-9 and
    -0.0; a larger frame (burst kernel + ragged rest) is held to NumPy's own cast of the same expression. Bit-exact."""
    g = golden("merge_float")
    v = g["frames_f64"]
    assert v.min() <= -3 / 255 and (v < 0).sum() >= 20
    neg = v[1] < 0
    _, _, idx = eng.linearize(dev(v[1]), None, g["icrf"], return_index=True)
    assert np.array_equal(host(idx)[neg], g["idx"][1][neg]) and set(np.unique(g["idx"][1][neg])) == {0, 253, 254, 255}
    rng = np.random.default_rng(12)
    big = rng.uniform(-0.02, 1.02, size=(67, 113, 3))
    big.reshape(-1)[:9] = [-0.5 / 255, -1.5 / 255, -2.5 / 255, -3.5 / 255, -0.0, 255.5 / 255, 256.5 / 255, -1 / 255, -4.0]
    want = np.around(big * 255).astype(np.uint8)                                  # measurand.py:503
    val, _, idx = eng.linearize(dev(big), None, g["icrf"], return_index=True)
    assert np.array_equal(host(idx), want)
    assert np.array_equal(host(val), np.take_along_axis(g["icrf"], want.reshape(-1, 3).astype(np.int64), axis=0).reshape(big.shape))
    # the float64 merge kernels compute the same index: a two-frame stack of the negative-bearing frame against the oracle
    fr = [big, np.clip(big * 2, -0.01, 1.0)]
    sd = [np.full(big.shape, 0.01), np.full(big.shape, 0.02)]
    ref = orc.merge(fr, [0.001, 0.002], g["icrf"], g["icrf_diff"], stds=sd)
    out = eng.merge([dev(f) for f in fr], [0.001, 0.002], g["icrf"], g["icrf_diff"], [dev(s_) for s_ in sd])
    close(host(out["val"]), ref["val"], F64_RTOL)
    close(host(out["std"]), ref["std"], STD_RTOL)
    out = eng.merge([dev(f) for f in fr], [0.001, 0.002], g["icrf"])
    close(host(out["val"]), ref["val"], F64_RTOL)


def test_linearize_u8_and_single_channel_lut(eng, golden):
    g = golden("merge_std")
    f = g["frames"][1]
    val, std = eng.linearize(dev(f), dev(g["stds"][1]), g["icrf"], g["icrf_diff"])
    assert np.array_equal(host(val), g["lin1_val"])
    close(host(std), g["lin1_std"], 1e-15)
    # no ICRF_diff -> no std (measurand.py:498-500)
    val2, std2 = eng.linearize(dev(f), dev(g["stds"][1]), g["icrf"])
    assert std2 is None and np.array_equal(host(val2), g["lin1_val"])
    # 1-D LUT applied to every channel (ImageSet.calculate_numerical_STD, image_set.py:383)
    lut = np.linspace(0.001, 0.02, 256)
    val3, _ = eng.linearize(dev(f), None, lut)
    assert np.array_equal(host(val3), lut[f])
    # float values on the DN grid give the same index as the DN itself
    val4, _, idx4 = eng.linearize(dev(f.astype(np.float64) / 255), None, g["icrf"], return_index=True)
    assert np.array_equal(host(idx4), f) and np.array_equal(host(val4), g["lin1_val"])


def test_gaussian_weight(eng, golden):
    g = golden("merge_float")
    w, dw = eng.gaussian_weight(dev(g["frames_f64"][0]))
    close(host(w), g["w0"], 1e-14)
    close(host(dw), g["dw0"], 1e-14)
    r = golden("merge_ramp")
    dn = np.arange(256, dtype=np.uint8)
    w, dw = eng.gaussian_weight(dev(dn))
    assert np.array_equal(host(w), r["w_lut"]) and np.array_equal(host(dw), r["dw_lut"])   # LUT path: bit-identical
    assert np.array_equal(host(eng.u8_to_unit(dev(dn))), dn.astype(np.float64) / 255)
    # analytic weights of a frame that is several 512-element burst chunks plus a ragged rest, and of a view that is not 16-byte aligned
    # (element-by-element kernel): the same bits from both kernels, 1e-14 from np.e ** x
    rng = np.random.default_rng(6)
    v = rng.random((37, 71, 3)) * 1.2 - 0.1
    w, dw = eng.gaussian_weight(dev(v))
    ow, odw = orc.gaussian_weight(v)
    close(host(w), ow, 1e-14)
    close(host(dw), odw, 1e-14)
    big = torch.zeros(v.size + 1, dtype=torch.float64, device="cuda")
    view = big[1:].view(v.shape)
    view.copy_(dev(v))
    w2, dw2 = eng.gaussian_weight(view)
    assert torch.equal(w2, w) and torch.equal(dw2, dw)


def test_hot_pixel_filter_standalone(eng):
    rng = np.random.default_rng(11)
    for (h, w, k) in ((9, 13, 3), (7, 5, 5), (3, 4, 3), (2, 2, 3)):
        x = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
        xs = rng.random((h, w, 3))
        dark = rng.integers(0, 40, size=(h, w, 3)).astype(np.uint8)
        thr = 0.1
        ref = orc.hot_pixel_filter(x.astype(np.float64), orc.unit_from_u8(dark), thr, k)
        out = eng.hot_pixel_filter(dev(x), dev(dark), thr, k)
        assert np.array_equal(host(out).astype(np.float64), ref)
        refs = orc.hot_pixel_filter(xs, orc.unit_from_u8(dark), thr, k)
        outs = eng.hot_pixel_filter(dev(xs), dev(orc.unit_from_u8(dark)), thr, k)
        assert np.array_equal(host(outs), refs)


@pytest.mark.parametrize("density", [1e-3, 0.3, 1.0])
@pytest.mark.parametrize("k", [3, 5])
def test_hot_pixel_filter_burst_and_tail(eng, density, k):
    """The standalone filter's burst kernel (whole 4 KB-per-wave spans, hot elements patched by their own lane) plus the chunk-by-chunk
    kernel for the ragged rest: 67 x 129 x 3 = 25 929 elements are 6 spans + a tail for uint8 data and 50 spans + a tail for float64
    data; sparse, dense and all-hot uint8 maps; a misaligned view (16-byte alignment lost: everything through the old kernel)."""
    rng = np.random.default_rng(int(1000 * density) + k)
    h, w = 67, 129
    x = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    xs = rng.random((h, w, 3))
    dark = rng.integers(0, 20, size=(h, w, 3)).astype(np.uint8)
    dark[rng.random((h, w, 3)) < density] = 200
    thr = 0.1
    dmap = orc.unit_from_u8(dark)
    ref = orc.hot_pixel_filter(x.astype(np.float64), dmap, thr, k)
    assert np.array_equal(host(eng.hot_pixel_filter(dev(x), dev(dark), thr, k)).astype(np.float64), ref)
    refs = orc.hot_pixel_filter(xs, dmap, thr, k)
    assert np.array_equal(host(eng.hot_pixel_filter(dev(xs), dev(dark), thr, k)), refs)            # float64 data, uint8 map: burst kernel
    big = torch.zeros(x.size + 16, dtype=torch.uint8, device="cuda")
    view = big[3:3 + x.size].view(h, w, 3)
    view.copy_(dev(x))
    assert not view.data_ptr() % 16 == 0
    assert np.array_equal(host(eng.hot_pixel_filter(view, dev(dark), thr, k)).astype(np.float64), ref)


# ------------------------------------------------------------------ seeded stacks vs the oracle
@pytest.mark.parametrize("n,h,w", [(1, 5, 7), (2, 16, 16), (7, 33, 29), (7, 64, 128), (15, 24, 40), (16, 9, 11),
                                   (17, 8, 8), (32, 4, 6), (24, 40, 33), (32, 31, 50), (17, 64, 128), (18, 40, 33), (19, 31, 50), (20, 64, 128),
                                   (21, 40, 33)])
@pytest.mark.parametrize("with_std", [False, True])
def test_merge_vs_oracle_sizes(eng, n, h, w, with_std):
    """Covers the fast kernel (N <= 20, whole 256/512-element groups), its tail, the run-time-N kernel (N > 20) and
    the generic kernel (tiny images, tails)."""
    frames, stds, t = orc.synthetic_stack(100 + n, n, h, w, with_std=with_std)
    icrf, diff = orc.synthetic_icrf()
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    out = eng.merge([dev(f) for f in frames], t, icrf, diff, [dev(s) for s in stds] if with_std else None)
    close(host(out["val"]), ref["val"], VAL_RTOL)
    if with_std:
        close(host(out["std"]), ref["std"], STD_RTOL)


def test_merge_uniform_random_dn(eng):
    """Uniform random DNs: worst case for LDS bank conflicts, every table entry exercised."""
    rng = np.random.default_rng(5)
    n, h, w = 7, 96, 128
    frames = [rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(n)]
    t = 1e-3 * 2.0 ** np.arange(n)
    icrf, diff = orc.synthetic_icrf()
    ref = orc.merge(frames, t, icrf, diff)
    out = eng.merge([dev(f) for f in frames], t, icrf, diff)
    close(host(out["val"]), ref["val"], VAL_RTOL)


def test_merge_row_tiles_with_halo(eng):
    """row tiles with a median halo reproduce the whole-image result exactly."""
    rng = np.random.default_rng(9)
    n, h, w = 5, 37, 24
    frames, stds, t = orc.synthetic_stack(9, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    dark = rng.integers(0, 30, size=(h, w, 3)).astype(np.uint8)
    dark[0, 0, 0] = 255
    dark[h - 1, w - 1, 2] = 255
    darks = [None, None, dark, dark, dark]
    mins = [256, 256, 20, 20, 20]
    whole = eng.merge([dev(f) for f in frames], t, icrf, diff, [dev(s) for s in stds],
                      darks=[None if d is None else dev(d) for d in darks], dark_min=mins, median_k=3)
    ref = orc.merge(frames, t, icrf, diff, stds=stds,
                    darks=[None if d is None else orc.unit_from_u8(d) for d in darks], dark_threshold=19.5 / 255, median_k=3)
    close(host(whole["val"]), ref["val"], VAL_RTOL)
    close(host(whole["std"]), ref["std"], STD_RTOL)
    bounds = [0, 10, 11, 25, 37]
    val = np.empty((h, w, 3))
    std = np.empty((h, w, 3))
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        b0, b1 = max(0, r0 - 1), min(h, r1 + 1)
        sl = slice(b0, b1)
        out = eng.merge([dev(f[sl]) for f in frames], t, icrf, diff, [dev(s[sl]) for s in stds],
                        darks=[None if d is None else dev(d[sl]) for d in darks], dark_min=mins, median_k=3,
                        height=h, row0=r0, rows=r1 - r0, buf_row0=b0)
        val[r0:r1] = host(out["val"])
        std[r0:r1] = host(out["std"])
    assert np.array_equal(val, host(whole["val"]))
    assert np.array_equal(std, host(whole["std"]))


@pytest.mark.parametrize("n,k,f64", [(15, 5, False), (9, 7, False), (7, 3, True), (32, 3, False), (3, 5, True), (20, 3, True), (16, 3, True)])
def test_hot_pixel_fixup_batches(eng, n, k, f64):
    """The hot-pixel fix-up takes floor(64 / k^2) frames' medians per batch (k = 3: 7, k = 5: 2, k = 7: 1):
    cover several batches, float64 frames, N up to 32, adjacent hot pixels, image corners and edges."""
    rng = np.random.default_rng(100 * n + k)
    h, w = 13, 11
    frames, stds, t = orc.synthetic_stack(50 + n, n, h, w, with_std=True)
    if f64:
        frames = [orc.unit_from_u8(f) + rng.random(f.shape) * 1e-3 for f in frames]
    icrf, diff = orc.synthetic_icrf()
    darks, mins, odarks = [], [], []
    for i in range(n):
        if i % 3 == 1:
            darks.append(None); mins.append(256); odarks.append(None)
            continue
        d = rng.integers(0, 10, size=(h, w, 3)).astype(np.uint8)
        d[rng.random((h, w, 3)) < 0.06] = 200
        d[0, 0, i % 3] = 255; d[h - 1, w - 1, (i + 1) % 3] = 255; d[0, w - 1, 0] = 255; d[h // 2, 0, 1] = 255
        darks.append(dev(d)); mins.append(100); odarks.append(orc.unit_from_u8(d))
    ref = orc.merge(frames, t, icrf, diff, stds=stds, darks=odarks, dark_threshold=99.5 / 255, median_k=k)
    out = eng.merge([dev(f) for f in frames], t, icrf, diff, [dev(s) for s in stds], darks=darks, dark_min=mins, median_k=k)
    close(host(out["val"]), ref["val"], F64_RTOL if f64 else VAL_RTOL)
    close(host(out["std"]), ref["std"], STD_RTOL)
    # val-only with the same hot maps
    ref = orc.merge(frames, t, icrf, diff, darks=odarks, dark_threshold=99.5 / 255, median_k=k)
    out = eng.merge([dev(f) for f in frames], t, icrf, diff, darks=darks, dark_min=mins, median_k=k, want_sum_w=True)
    close(host(out["val"]), ref["val"], F64_RTOL if f64 else VAL_RTOL)
    close(host(out["sum_w"]), ref["S"], 1e-13)
    # the workspace-free pass (one hot element per wave at a time) and the queue pass (one per lane) give the same bits
    for kw in (dict(stds=[dev(s) for s in stds]), dict(want_sum_w=True)):
        a_ = eng.merge([dev(f) for f in frames], t, icrf, diff, darks=darks, dark_min=mins, median_k=k, hot_queue=True, **kw)
        b_ = eng.merge([dev(f) for f in frames], t, icrf, diff, darks=darks, dark_min=mins, median_k=k, hot_queue=False, **kw)
        for key in a_:
            assert torch.equal(a_[key], b_[key]), key


@pytest.mark.parametrize("density", [0.0, 1e-3, 0.05, 0.3, 1.0])
@pytest.mark.parametrize("shared_map", [True, False])
def test_hot_pixel_queue_at_every_density(eng, density, shared_map):
    """The queue-driven hot-pixel pass (scan -> one queued element per lane) against the oracle and against the workspace-free
    pass, from an empty map to one where EVERY element is hot (the queue holds a quarter of the elements: 0.3 and 1.0 overflow
    it and must fall back, bit-identically), one map shared by all frames and a different map per frame; 137 x 61 x 3 elements
    are not a multiple of the 1024-element span a wave scans."""
    n, h, w = 5, 137, 61
    rng = np.random.default_rng(int(density * 1000) + 7 * shared_map)
    frames, stds, t = orc.synthetic_stack(77, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    def one_map():
        d = rng.integers(0, 10, size=(h, w, 3)).astype(np.uint8)
        d[rng.random((h, w, 3)) < density] = 200
        return d
    maps = [one_map()] * n if shared_map else [one_map() for _ in range(n)]
    dk = [dev(m) for m in maps]
    if shared_map:
        dk = [dk[0]] * n
    ref = orc.merge(frames, t, icrf, diff, stds=stds, darks=[orc.unit_from_u8(m) for m in maps], dark_threshold=99.5 / 255, median_k=3)
    fr, sd = [dev(f) for f in frames], [dev(s) for s in stds]
    plan = eng.plan_merge(fr, t, icrf, diff, sd, darks=dk, dark_min=[100] * n, median_k=3)
    assert "merge_scan_hot + merge_patch_hot" in plan.kernels
    plan.launch()
    close(host(plan.outputs["val"]), ref["val"], VAL_RTOL)
    close(host(plan.outputs["std"]), ref["std"], STD_RTOL)
    old = eng.plan_merge(fr, t, icrf, diff, sd, darks=dk, dark_min=[100] * n, median_k=3, hot_queue=False)
    assert "merge_scan_hot" not in old.kernels and "merge_fixup_hot" in old.kernels
    old.launch()
    assert torch.equal(plan.outputs["val"], old.outputs["val"]) and torch.equal(plan.outputs["std"], old.outputs["std"])
    # launching the same plan again reuses its workspace (the counters are reset on the stream)
    plan.outputs["val"].zero_()
    plan.launch()
    assert torch.equal(plan.outputs["val"], old.outputs["val"])


def test_hot_pixel_queue_smallest_workspace(eng):
    """Any workspace from hm_merge_hot_workspace_min_bytes() up is legal: a one-entry queue overflows on the second hot element and
    the patch kernel goes over the whole tile instead; one byte less selects the workspace-free pass. Same bits every way."""
    from camera_linearity_amd import _native as nat
    n, h, w = 3, 40, 33
    frames, stds, t = orc.synthetic_stack(78, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(78)
    d = (rng.random((h, w, 3)) < 0.02).astype(np.uint8) * 200
    fr, sd, dk = [dev(f) for f in frames], [dev(s) for s in stds], dev(d)
    plan = eng.plan_merge(fr, t, icrf, diff, sd, darks=[dk] * n, dark_min=[100] * n, median_k=3)
    least = int(nat.lib.hm_merge_hot_workspace_min_bytes(h * w * 3))
    small = torch.zeros(least + 16, dtype=torch.uint8, device="cuda")
    plan.args.hot_workspace, plan.args.hot_workspace_bytes = small.data_ptr(), least
    assert "merge_scan_hot" in plan.kernels
    plan.launch()
    torch.cuda.synchronize()
    words = small.cpu().numpy()[:16].view(np.uint32)
    assert words[0] >= 1 and words[1] == 1                 # something was queued, then the queue overflowed
    plan.args.hot_workspace_bytes = least - 1
    assert "merge_scan_hot" not in plan.kernels and "merge_fixup_hot" in plan.kernels
    plan.args.hot_workspace_bytes = least
    old = eng.plan_merge(fr, t, icrf, diff, sd, darks=[dk] * n, dark_min=[100] * n, median_k=3, hot_queue=False)
    old.launch()
    assert torch.equal(plan.outputs["val"], old.outputs["val"]) and torch.equal(plan.outputs["std"], old.outputs["std"])


@pytest.mark.parametrize("n", [7, 8, 17, 18, 19, 20, 21, 32])          # (8 + std + flat + sum of weights: the instantiation that spilled 20 B/lane in round 3)
@pytest.mark.parametrize("with_std", [False, True])
def test_generic_kernel_bit_identical_to_fast_kernel(eng, with_std, n):
    """variant < 0 forces merge_generic; it must reproduce merge_u8_fast (N <= 20) and merge_u8_loop (run-time frame
    count, 20 < N <= 32) bit for bit (shared operation sequence), including flat field and sum-of-weights output."""
    h, w = 96, 130
    frames, stds, t = orc.synthetic_stack(77, n, h, w, with_std=with_std)
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(3)
    flat = dev(rng.integers(180, 230, size=(h, w, 3)).astype(np.uint8))
    flat_std = dev(np.full((h, w, 3), 0.002))
    fr = [dev(f) for f in frames]
    sd = [dev(s) for s in stds] if with_std else None
    kw = dict(flat=flat, ff_mean=[0.8, 0.81, 0.79], want_sum_w=True)
    if with_std:
        kw.update(flat_std=flat_std, ff_std_mean=[0.002] * 3)
    a = eng.merge(fr, t, icrf, diff, sd, variant=0, **kw)
    b = eng.merge(fr, t, icrf, diff, sd, variant=-1, **kw)
    for key in a:
        assert torch.equal(a[key], b[key]), key
    # and without the extras (the plain instantiations)
    a = eng.merge(fr, t, icrf, diff, sd, variant=0)
    b = eng.merge(fr, t, icrf, diff, sd, variant=-1)
    for key in a:
        assert torch.equal(a[key], b[key]), key
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    close(host(a["val"]), ref["val"], VAL_RTOL)
    # and without the extras (the plain instantiations)
    a = eng.merge(fr, t, icrf, diff, sd, variant=0)
    b = eng.merge(fr, t, icrf, diff, sd, variant=-1)
    for key in a:
        assert torch.equal(a[key], b[key]), key
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    close(host(a["val"]), ref["val"], VAL_RTOL)


@pytest.mark.parametrize("C", [1, 2, 4])
@pytest.mark.parametrize("with_std", [False, True])
@pytest.mark.parametrize("n,h,w", [(4, 9, 7), (5, 40, 52), (20, 33, 31)])
def test_merge_other_channel_counts(eng, C, with_std, n, h, w):
    """C != 3 (HM_MAX_CHANNELS = 4): whole 128-element groups stream through merge_u8_loop<C>, the rest through the
    generic kernel; both against the oracle, and bit-identical to the all-generic result, with flat field and sum of
    weights."""
    frames, stds, t = orc.synthetic_stack(300 + C, n, h, w, c=C, with_std=with_std)
    icrf = np.stack([np.linspace(0, 1, 256) ** (1.5 + 0.2 * k) for k in range(C)], axis=1)
    diff = orc.icrf_derivative(icrf)
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    fr = [dev(f) for f in frames]
    sd = [dev(s) for s in stds] if with_std else None
    out = eng.merge(fr, t, icrf, diff, sd)
    close(host(out["val"]), ref["val"], VAL_RTOL)
    if with_std:
        close(host(out["std"]), ref["std"], STD_RTOL)
    rng = np.random.default_rng(C)
    kw = dict(flat=dev(rng.integers(180, 230, size=(h, w, C)).astype(np.uint8)), ff_mean=[0.8, 0.81, 0.79, 0.82][:C], want_sum_w=True)
    if with_std:
        kw.update(flat_std=dev(np.full((h, w, C), 0.002)), ff_std_mean=[0.002] * C)
    a_ = eng.merge(fr, t, icrf, diff, sd, variant=0, **kw)
    b_ = eng.merge(fr, t, icrf, diff, sd, variant=-1, **kw)
    for key in a_:
        assert torch.equal(a_[key], b_[key]), key


def test_merge_unaligned_tile_and_single_row(eng):
    """Row tiles whose byte offset is odd (W*C odd) cannot use the 2-byte loads of the fast kernel: the generic kernel
    takes over and the result is still bit-identical to the whole-image launch. Also rows == 0 and a 1 x 1 image."""
    n, h, w = 3, 11, 5                      # W*C = 15 bytes per row: every other row starts at an odd address
    frames, stds, t = orc.synthetic_stack(41, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    fr, sd = [dev(f) for f in frames], [dev(s) for s in stds]
    whole = eng.merge(fr, t, icrf, diff, sd)
    ref = orc.merge(frames, t, icrf, diff, stds=stds)
    close(host(whole["val"]), ref["val"], VAL_RTOL)
    for r0, r1 in ((1, 4), (3, 11), (0, 1), (10, 11)):
        out = eng.merge(fr, t, icrf, diff, sd, height=h, row0=r0, rows=r1 - r0, buf_row0=0)
        assert torch.equal(out["val"], whole["val"][r0:r1]) and torch.equal(out["std"], whole["std"][r0:r1])
    empty = eng.merge(fr, t, icrf, diff, sd, height=h, row0=4, rows=0, buf_row0=0)
    assert tuple(empty["val"].shape) == (0, w, 3)
    one = eng.merge([dev(f[:1, :1]) for f in frames], t, icrf, diff)
    close(host(one["val"]), orc.merge([f[:1, :1] for f in frames], t, icrf, diff)["val"], VAL_RTOL)


def test_row_tile_sharding_helper_on_device(eng):
    """parallel.merge_row_tile (what each rank of a row-sharded merge runs): the tiles of world sizes 1, 3 and 8,
    computed one after the other on this GPU and concatenated, equal the whole-image merge bit for bit
    (hot-pixel halo rows and flat-field rows included)."""
    from camera_linearity_amd import parallel
    rng = np.random.default_rng(12)
    n, h, w = 5, 41, 18
    frames, stds, t = orc.synthetic_stack(12, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    dark = rng.integers(0, 30, size=(h, w, 3)).astype(np.uint8)
    dark[rng.random((h, w, 3)) < 0.02] = 250
    flat = rng.integers(180, 230, size=(h, w, 3)).astype(np.uint8)
    flat_std = np.full((h, w, 3), 0.002)
    darks = [None, dark, dark, None, dark]
    mins = [256, 40, 40, 256, 40]
    kw = dict(stds_host=stds, darks_host=darks, dark_min=mins, median_k=3, flat_host=flat, flat_std_host=flat_std,
              ff_mean=[0.8, 0.79, 0.81], ff_std_mean=[0.002] * 3)
    _, _, whole_val, whole_std = parallel.merge_row_tile(frames, t, icrf, diff, rank=0, world_size=1, **kw)
    ref = orc.merge(frames, t, icrf, diff, stds=stds, darks=[None if d is None else orc.unit_from_u8(d) for d in darks],
                    dark_threshold=39.5 / 255, median_k=3, flat=orc.unit_from_u8(flat), flat_std=flat_std,
                    ff_mean=np.array([0.8, 0.79, 0.81]), ff_std_mean=np.array([0.002] * 3))
    close(whole_val, ref["val_ff"], VAL_RTOL)
    close(whole_std, ref["std_ff"], STD_RTOL)
    for world in (3, 8):
        vals, sds = [], []
        for r in range(world):
            r0, r1, v, s = parallel.merge_row_tile(frames, t, icrf, diff, rank=r, world_size=world, **kw)
            assert (r0, r1) == parallel.row_tile_bounds(h, world)[r] and v.shape[0] == r1 - r0
            vals.append(v); sds.append(s)
        assert np.array_equal(np.concatenate(vals), whole_val)
        assert np.array_equal(np.concatenate(sds), whole_std)


def test_merge_argument_errors(eng):
    f = dev(np.zeros((4, 4, 3), np.uint8))
    icrf, diff = orc.synthetic_icrf()
    with pytest.raises(ValueError):
        eng.merge([f], [0.0], icrf)                               # non-positive exposure
    with pytest.raises(ValueError):
        eng.merge([f, f], [1.0], icrf)                            # exposures length
    with pytest.raises(ValueError):
        eng.merge([f], [1.0], icrf[:, :2])                        # ICRF shape
    with pytest.raises(ValueError):
        eng.merge([f], [1.0], icrf, None, [dev(np.zeros((4, 4, 3)))])   # std without ICRF_diff
    assert eng.merge([f] * 33, [1.0] * 33, icrf)["val"].shape == (4, 4, 3)   # > HM_MAX_FRAMES: chunked, no longer an error
    with pytest.raises(RuntimeError):
        eng.merge([torch.zeros((4, 4, 3), dtype=torch.uint8)], [1.0], icrf)   # host tensor: no CPU fallback
    d = dev(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(ValueError):                               # halo missing for the median
        eng.merge([f], [1.0], icrf, darks=[d], dark_min=[1], median_k=3, height=8, row0=2, rows=4, buf_row0=2)


def test_sum_of_weights(eng, golden):
    g = golden("merge_std")
    S, S2 = eng.sum_of_weights([dev(f) for f in g["frames"]])
    close(host(S), g["S"], 1e-15 * 8)
    close(host(S2), g["S"] ** 2, 1e-14)


def test_merge_captured_in_hip_graph(eng):
    """hm_merge allocates nothing and does not synchronise (include/hdrmerge.h conventions), so a batch of small merges -
    the launch-bound regime of config 1 - can be captured once into a hipGraph and replayed: same bits as eager launches,
    also after the inputs change in place (the graph holds pointers, not data)."""
    icrf, diff = orc.synthetic_icrf()
    stacks = [orc.synthetic_stack(20 + k, 3, 64, 64, with_std=True) for k in range(8)]
    flat = np.clip(np.around(255 * (0.8 + 0.05 * np.random.default_rng(0).random((64, 64, 3)))), 1, 255).astype(np.uint8)
    dark = (np.random.default_rng(1).random((64, 64, 3)) < 0.01).astype(np.uint8) * 200
    plans, inputs = [], []
    for frames, stds, t in stacks:
        fr = [torch.as_tensor(f, device="cuda") for f in frames]
        inputs.append(fr)
        sd = [torch.as_tensor(s, device="cuda") for s in stds]
        plans.append(eng.plan_merge(fr, t, icrf, diff, sd, darks=[torch.as_tensor(dark, device="cuda")] * 3, dark_min=[13] * 3,
                                    median_k=3))
    for p in plans:                                   # eager reference (also warms the library's cached device queries)
        p.launch()
    torch.cuda.synchronize()
    eager = [(p.outputs["val"].clone(), p.outputs["std"].clone()) for p in plans]
    for p in plans:
        p.outputs["val"].zero_(); p.outputs["std"].zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p in plans:
            p.launch()
    g.replay()
    torch.cuda.synchronize()
    for p, (v, s) in zip(plans, eager):
        assert torch.equal(p.outputs["val"], v) and torch.equal(p.outputs["std"], s)
    # change one stack's frames in place and replay: the graph recomputes from the new bytes
    for dst, src in zip(inputs[0], orc.synthetic_stack(99, 3, 64, 64)[0]):
        dst.copy_(torch.as_tensor(src, device="cuda"))
    plans[0].launch()
    torch.cuda.synchronize()
    want = plans[0].outputs["val"].clone()
    assert not torch.equal(want, eager[0][0])
    plans[0].outputs["val"].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(plans[0].outputs["val"], want) and torch.equal(plans[1].outputs["val"], eager[1][0])


@pytest.mark.parametrize("n,C", [(3, 3), (7, 3), (20, 3), (5, 1), (6, 4)])
@pytest.mark.parametrize("with_std", [False, True])
def test_f64_streaming_kernel(eng, with_std, n, C):
    """float64 frames (64-bit mode): merge_f64_loop against the oracle (analytic weights, computed index incl. .5 ties and
    values off the DN grid) and bit-identical to merge_generic (variant < 0), with and without flat field / sum of weights."""
    h, w = 48, 70
    rng = np.random.default_rng(n * 10 + C)
    v = [rng.random((h, w, C)) for _ in range(n)]
    v[0].reshape(-1)[:64] = (np.arange(64) + 0.5) / 255                  # exact .5 ties where representable
    v[n - 1].reshape(-1)[:16] = 1.0 + rng.random(16) * 0.004             # rounds to 255 or wraps to 0
    stds = [0.01 * (1 + rng.random((h, w, C))) for _ in range(n)] if with_std else None
    t = 1e-3 * 1.6 ** np.arange(n)
    icrf = np.stack([np.linspace(0, 1, 256) ** (1.6 + 0.2 * k) for k in range(C)], axis=1)
    diff = orc.icrf_derivative(icrf)
    ref = orc.merge(v, t, icrf, diff, stds=stds)
    fr = [dev(x) for x in v]
    sd = [dev(s) for s in stds] if with_std else None
    a = eng.merge(fr, t, icrf, diff, sd, variant=0)
    b = eng.merge(fr, t, icrf, diff, sd, variant=-1)
    close(host(a["val"]), ref["val"], 1e-11)
    if with_std:
        close(host(a["std"]), ref["std"], STD_RTOL)
    for key in a:
        assert torch.equal(a[key], b[key]), key
    kw = dict(flat=dev(rng.integers(180, 230, size=(h, w, C)).astype(np.uint8)), ff_mean=[0.8, 0.81, 0.79, 0.82][:C], want_sum_w=True)
    if with_std:
        kw.update(flat_std=dev(np.full((h, w, C), 0.002)), ff_std_mean=[0.002] * C)
    a = eng.merge(fr, t, icrf, diff, sd, variant=0, **kw)
    b = eng.merge(fr, t, icrf, diff, sd, variant=-1, **kw)
    for key in a:
        assert torch.equal(a[key], b[key]), key


def test_merge_special_values(eng):
    """NaN / inf in the std frames, an all-dark and an all-saturated pixel column, an ICRF with values above 1 and a
    non-monotonic ICRF: the kernels propagate what NumPy propagates (same NaN positions, same finite values)."""
    n, h, w = 5, 16, 40
    frames, stds, t = orc.synthetic_stack(55, n, h, w, with_std=True)
    for f in frames:
        f[:, 0] = 0          # all-dark column: radiance exactly 0 with ICRF[0] = 0
        f[:, 1] = 255        # all-saturated column
    stds[1][2, 3, 1] = np.nan
    stds[3][4, 5, 2] = np.inf
    stds[0][6, 7, 0] = 0.0
    icrf, _ = orc.synthetic_icrf()
    icrf = icrf * 1.7
    icrf[100:110, 1] = icrf[100:110, 1][::-1]           # locally decreasing
    diff = orc.icrf_derivative(icrf)
    with np.errstate(all="ignore"):
        ref = orc.merge(frames, t, icrf, diff, stds=stds)
    out = eng.merge([dev(f) for f in frames], t, icrf, diff, [dev(s) for s in stds])
    val, std = host(out["val"]), host(out["std"])
    assert np.array_equal(np.isnan(std), np.isnan(ref["std"])) and np.array_equal(np.isinf(std), np.isinf(ref["std"]))
    assert np.isnan(std[2, 3, 1]) and not np.isfinite(std[4, 5, 2])
    ok = np.isfinite(ref["std"])
    close(std[ok], ref["std"][ok], STD_RTOL)
    close(val, ref["val"], VAL_RTOL)
    assert (val[:, 0] == 0).all()


def test_merge_randomised_configurations(eng):
    """120 seeded random configurations (frame count 1..32, ragged sizes, C in 1..4, with / without std, dark maps with
    3x3 / 5x5 medians, flat field, sum-of-weights output, row tiles with halo) against the oracle."""
    master = np.random.default_rng(2024)
    for case in range(120):
        rng = np.random.default_rng(master.integers(0, 2 ** 31))
        n = int(rng.choice([1, 2, 3, 5, 7, 9, 15, 16, 17, 23, 32]))
        C = int(rng.choice([3, 3, 3, 1, 2, 4]))
        h, w = int(rng.integers(6, 48)), int(rng.integers(5, 70))
        with_std = bool(rng.integers(0, 2))
        use_hot = bool(rng.integers(0, 2))
        use_flat = bool(rng.integers(0, 2))
        k = int(rng.choice([3, 5]))
        frames, stds, t = orc.synthetic_stack(int(rng.integers(0, 10 ** 6)), n, h, w, c=C, with_std=with_std)
        icrf = np.stack([np.linspace(0, 1, 256) ** (1.4 + 0.3 * c) for c in range(C)], axis=1)
        diff = orc.icrf_derivative(icrf)
        thr = 0.05
        darks_dn = None
        if use_hot:
            darks_dn = []
            for i in range(n):
                if rng.random() < 0.3:
                    darks_dn.append(None)
                else:
                    d = (rng.random((h, w, C)) < 0.03).astype(np.uint8) * rng.integers(20, 255, (h, w, C)).astype(np.uint8)
                    darks_dn.append(d)
        flat = rng.integers(150, 240, (h, w, C)).astype(np.uint8) if use_flat else None
        flat_std = np.full((h, w, C), 0.002) + 0.001 * rng.random((h, w, C)) if use_flat else None
        ff_mean = list(0.7 + 0.1 * rng.random(C)) if use_flat else None
        ff_std_mean = list(0.002 + 0.001 * rng.random(C)) if use_flat else None
        ref = orc.merge(frames, t, icrf, diff if with_std else None, stds=stds,
                        darks=None if darks_dn is None else [None if d is None else orc.unit_from_u8(d) for d in darks_dn],
                        dark_threshold=thr, median_k=k,
                        flat=orc.unit_from_u8(flat) if (use_flat and with_std) else None, flat_std=flat_std if with_std else None,
                        ff_mean=np.array(ff_mean) if (use_flat and with_std) else None,
                        ff_std_mean=np.array(ff_std_mean) if (use_flat and with_std) else None)
        if use_flat and not with_std:                     # val-only flat field = measurand.py:602 alone
            ref["val_ff"] = (ref["val"] / orc.unit_from_u8(flat)) * np.array(ff_mean)
        kw = {}
        if use_hot:
            kw.update(darks=[None if d is None else dev(d) for d in darks_dn], dark_min=[eng.dark_min_dn(1.0, thr)] * n, median_k=k)
        if use_flat:
            kw.update(flat=dev(flat), ff_mean=ff_mean)
            if with_std:
                kw.update(flat_std=dev(flat_std), ff_std_mean=ff_std_mean)
        fr = [dev(f) for f in frames]
        sd = [dev(s) for s in stds] if with_std else None
        out = eng.merge(fr, t, icrf, diff if with_std else None, sd, want_sum_w=True, **kw)
        tag = f"case {case}: n={n} C={C} {h}x{w} std={with_std} hot={use_hot} flat={use_flat} k={k}"
        want_val = ref["val_ff"] if use_flat else ref["val"]
        np.testing.assert_allclose(host(out["val"]), want_val, rtol=VAL_RTOL, atol=0, err_msg=tag)
        np.testing.assert_allclose(host(out["sum_w"]), ref["S"], rtol=1e-13, atol=0, err_msg=tag)
        if with_std:
            np.testing.assert_allclose(host(out["std"]), ref["std_ff"] if use_flat else ref["std"], rtol=STD_RTOL, atol=0, err_msg=tag)
        # the same image as two row tiles with the median halo: bit-identical to the whole-image result
        if h >= 8:
            cut = int(rng.integers(2, h - 2))
            r = k // 2
            parts = []
            for (r0, r1) in ((0, cut), (cut, h)):
                b0, b1 = max(0, r0 - r), min(h, r1 + r)
                kw_t = {}
                if use_hot:
                    kw_t.update(darks=[None if d is None else dev(d[b0:b1]) for d in darks_dn], dark_min=[eng.dark_min_dn(1.0, thr)] * n, median_k=k)
                if use_flat:
                    kw_t.update(flat=dev(flat[r0:r1]), ff_mean=ff_mean)
                    if with_std:
                        kw_t.update(flat_std=dev(flat_std[r0:r1]), ff_std_mean=ff_std_mean)
                o = eng.merge([dev(f[b0:b1]) for f in frames], t, icrf, diff if with_std else None,
                              [dev(s[b0:b1]) for s in stds] if with_std else None, height=h, row0=r0, rows=r1 - r0, buf_row0=b0, **kw_t)
                parts.append(o)
            whole = host(out["val"])
            tiled = np.concatenate([host(p["val"]) for p in parts], axis=0)
            assert np.array_equal(whole, tiled), tag + " (row tiles)"


def test_merge_full_size_properties(eng):
    """BASELINE config 2 / 3 size (7 x 4096 x 4096 x 3), checked through size-independent properties instead of the oracle:
    (1) exposure scaling: doubling every exposure halves radiance and uncertainty exactly (power of two, bit-exact);
    (2) tiling invariance: two row tiles reproduce the whole image bit for bit;
    (3) kernel invariance: the generic kernel reproduces the streaming kernel bit for bit (checksum of the raw bits);
    (4) a row band against the oracle (1e-12);
    (5) constant frames: a stack whose frames all hold one DN merges to ICRF[DN] * sum(w / t) / sum(w) everywhere."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
    n, H, W = 7, 4096, 4096
    frames, stds, t = synthetic_stack_device(11, n, H, W, device="cuda", with_std=True)
    icrf, diff = synthetic_icrf()
    base = eng.merge(frames, t, icrf, diff, stds)
    val, std = base["val"], base["std"]
    # (1)
    dbl = eng.merge(frames, [2 * x for x in t], icrf, diff, stds)
    assert torch.equal(dbl["val"], val * 0.5) and torch.equal(dbl["std"], std * 0.5)
    del dbl
    # (2)
    cut = 1777
    top = eng.merge([f[:cut] for f in frames], t, icrf, diff, [s[:cut] for s in stds], height=H, row0=0, rows=cut, buf_row0=0)
    bot = eng.merge([f[cut:] for f in frames], t, icrf, diff, [s[cut:] for s in stds], height=H, row0=cut, rows=H - cut, buf_row0=cut)
    assert torch.equal(top["val"], val[:cut]) and torch.equal(bot["val"], val[cut:])
    assert torch.equal(top["std"], std[:cut]) and torch.equal(bot["std"], std[cut:])
    del top, bot
    # (3)
    gen = eng.merge(frames, t, icrf, diff, stds, variant=-1)
    def checksum(x):
        return int(x.view(torch.int64).sum().item())
    assert checksum(gen["val"]) == checksum(val) and checksum(gen["std"]) == checksum(std)
    assert torch.equal(gen["val"], val)
    del gen
    # (4)
    rows = 64
    ref = orc.merge([f[:rows].cpu().numpy() for f in frames], t, icrf, diff, stds=[s[:rows].cpu().numpy() for s in stds])
    close(host(val[:rows]), ref["val"], VAL_RTOL)
    close(host(std[:rows]), ref["std"], STD_RTOL)
    # (5) val-only, constant frames
    dn = 97
    const = [torch.full((H, W, 3), dn, dtype=torch.uint8, device="cuda") for _ in range(n)]
    out = eng.merge(const, t, icrf)["val"]
    w = orc.gaussian_weight_lut()[0][dn]
    expect = np.array([icrf[dn, c] * sum(w / ti for ti in t) / (n * w) for c in range(3)])
    got = out.reshape(-1, 3)
    assert bool((got == got[0]).all())
    np.testing.assert_allclose(host(got[0]), expect, rtol=1e-14)


def test_config1_shape_identity_icrf(eng):
    """BASELINE configs[0] on the GPU path: 3 x 256 x 256 x 3, identity ICRF, against the oracle."""
    frames, _, t = orc.synthetic_stack(0, 3, 256, 256)
    icrf = np.stack([np.arange(256) / 255.0] * 3, axis=1)
    ref = orc.merge(frames, t, icrf)
    out = eng.merge([dev(f) for f in frames], t, icrf)
    close(host(out["val"]), ref["val"], VAL_RTOL)


def test_concurrent_streams(eng):
    """hm_merge is re-entrant and stream-ordered: four stacks merged concurrently on four HIP streams (launches interleaved from
    one host thread, hot-pixel fix-up passes included) give the results of serial launches."""
    icrf, diff = orc.synthetic_icrf()
    dark = dev((np.random.default_rng(2).random((96, 128, 3)) < 0.01).astype(np.uint8) * 220)
    plans, serial = [], []
    for k in range(4):
        frames, stds, t = orc.synthetic_stack(70 + k, 6, 96, 128, with_std=True)
        p = eng.plan_merge([dev(f) for f in frames], t, icrf, diff, [dev(s) for s in stds], darks=[dark] * 6, dark_min=[13] * 6, median_k=3)
        p.launch()
        torch.cuda.synchronize()
        serial.append((p.outputs["val"].clone(), p.outputs["std"].clone()))
        p.outputs["val"].zero_(); p.outputs["std"].zero_()
        plans.append(p)
    streams = [torch.cuda.Stream() for _ in plans]
    torch.cuda.synchronize()
    for rep in range(5):                                  # interleave launches across the streams
        for p, s in zip(plans, streams):
            p.launch(stream=s.cuda_stream)
    for s in streams:
        s.synchronize()
    for p, (v, sd) in zip(plans, serial):
        assert torch.equal(p.outputs["val"], v) and torch.equal(p.outputs["std"], sd)


def test_config3_full_size_with_corrections(eng):
    """BASELINE configs[2] at the size the metric is quoted on - 7 x 4096 x 4096 x 3 uint8 + float64 std + seven dark maps
    (hot-pixel filter) + flat field - through size-independent properties and a band against the oracle:
    (1) tiling invariance: two row tiles with a 3 x 3 median halo reproduce the whole image bit for bit;
    (2) kernel invariance: the generic kernel (+ fix-up pass) reproduces the streaming kernel bit for bit;
    (3) rows 0..63 (+ one halo row) against the oracle: val 1e-12, std 1e-9."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark
    n, H, W, thr = 7, 4096, 4096, 0.05
    frames, stds, t = synthetic_stack_device(13, n, H, W, device="cuda", with_std=True)
    icrf, diff = synthetic_icrf()
    flat, flat_std, dark0 = synthetic_flat_dark(13, H, W, device="cuda", hot_density=1e-4)
    darks = [dark0] + [synthetic_flat_dark(14 + i, H, W, device="cuda", hot_density=1e-4)[2] for i in range(n - 1)]   # seven DIFFERENT maps
    x0, x1, y0, y1 = eng.flat_roi_bounds(H, W, 0.2)
    ffm = eng.roi_mean(flat, x0, x1, y0, y1).cpu().numpy()
    ffs = eng.roi_mean(flat_std, x0, x1, y0, y1).cpu().numpy()
    dmin = [eng.dark_min_dn(1.0, thr)] * n
    kw = dict(dark_min=dmin, median_k=3, ff_mean=ffm, ff_std_mean=ffs)
    base = eng.plan_merge(frames, t, icrf, diff, stds, darks=darks, flat=flat, flat_std=flat_std, **kw)
    assert "merge_u8_fast_std" in base.kernels and "merge_scan_hot + merge_patch_hot" in base.kernels
    base.launch()
    val, std = base.outputs["val"], base.outputs["std"]
    assert bool(torch.isfinite(val).all()) and bool(torch.isfinite(std).all())
    # (1)
    cut = 1777
    top = eng.merge([f[:cut + 1] for f in frames], t, icrf, diff, [s[:cut + 1] for s in stds], darks=[d[:cut + 1] for d in darks],
                    flat=flat[:cut], flat_std=flat_std[:cut], height=H, row0=0, rows=cut, buf_row0=0, **kw)
    bot = eng.merge([f[cut - 1:] for f in frames], t, icrf, diff, [s[cut - 1:] for s in stds], darks=[d[cut - 1:] for d in darks],
                    flat=flat[cut:], flat_std=flat_std[cut:], height=H, row0=cut, rows=H - cut, buf_row0=cut - 1, **kw)
    assert torch.equal(top["val"], val[:cut]) and torch.equal(bot["val"], val[cut:])
    assert torch.equal(top["std"], std[:cut]) and torch.equal(bot["std"], std[cut:])
    del top, bot
    # (2)
    gen = eng.merge(frames, t, icrf, diff, stds, darks=darks, flat=flat, flat_std=flat_std, variant=-1, **kw)
    assert torch.equal(gen["val"], val) and torch.equal(gen["std"], std)
    del gen
    # (3)
    rows = 64
    h = lambda x: x[:rows + 1].cpu().numpy()                                 # noqa: E731
    fv = orc.unit_from_u8(h(flat))
    ref = orc.merge([h(f) for f in frames], t, icrf, diff, stds=[h(s) for s in stds], darks=[orc.unit_from_u8(h(d)) for d in darks],
                    dark_threshold=thr, median_k=3, flat=fv, flat_std=h(flat_std), ff_mean=ffm, ff_std_mean=ffs)
    hot_rows = sum(int((d[:rows] >= dmin[0]).sum()) for d in darks)
    assert hot_rows > 0                                                       # the band does exercise the fix-up pass
    close(host(val[:rows]), ref["val_ff"][:rows], VAL_RTOL)
    close(host(std[:rows]), ref["std_ff"][:rows], STD_RTOL)


@pytest.mark.parametrize("with_std", [False, True])
def test_config4_row_tile_shape(eng, with_std):
    """BASELINE configs[3]'s tile at the size the metric is quoted on: 15 frames, 1024 rows x 8192 columns x 3, the tile at
    row0 = 3072 of an 8192-row image with a 3 x 3 median halo (input rows 3071..4096). It must equal, bit for bit, rows
    3072..4095 of a 2048-row tile starting at 2048 merged from the same buffers (different tiling, different group phase),
    and a band of it must match the oracle."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark
    n, Himg, W, thr = 15, 8192, 8192, 0.05
    lo, hi = 2047, 4097                                                       # buffer rows: the 2-tile span + halo
    frames, stds, t = synthetic_stack_device(17, n, hi - lo, W, device="cuda", with_std=with_std)
    icrf, diff = synthetic_icrf()
    dark = synthetic_flat_dark(17, hi - lo, W, device="cuda", hot_density=1e-3)[2]
    darks = [dark if i % 2 else None for i in range(n)]
    dmin = [eng.dark_min_dn(1.0, thr) if i % 2 else 256 for i in range(n)]
    kw = dict(darks=darks, dark_min=dmin, median_k=3, height=Himg)
    big = eng.plan_merge(frames, t, icrf, diff if with_std else None, stds, row0=2048, rows=2048, buf_row0=lo, **kw)
    big.launch()
    cutv = lambda x: None if x is None else x[3071 - lo:]                    # noqa: E731  (views: contiguous row slices, no copies)
    tile = eng.plan_merge([cutv(f) for f in frames], t, icrf, diff if with_std else None, None if stds is None else [cutv(s) for s in stds],
                          darks=[cutv(d) for d in darks], dark_min=dmin, median_k=3, height=Himg, row0=3072, rows=1024, buf_row0=3071)
    assert ("merge_u8_fast_std<N=15" if with_std else "merge_u8_val3<N=15") in tile.kernels and "merge_patch_hot" in tile.kernels
    tile.launch()
    assert torch.equal(tile.outputs["val"], big.outputs["val"][1024:])
    if with_std:
        assert torch.equal(tile.outputs["std"], big.outputs["std"][1024:])
    rows = 24
    h = lambda x: x[3071 - lo:3071 - lo + rows + 2].cpu().numpy()            # noqa: E731  (image rows 3071 .. 3072 + rows)
    dv = orc.unit_from_u8(h(dark))
    ref = orc.merge([h(f) for f in frames], t, icrf, diff if with_std else None, stds=None if stds is None else [h(s) for s in stds],
                    darks=[dv if i % 2 else None for i in range(n)], dark_threshold=thr, median_k=3)
    close(host(tile.outputs["val"][:rows]), ref["val"][1:rows + 1], VAL_RTOL)
    if with_std:
        close(host(tile.outputs["std"][:rows]), ref["std"][1:rows + 1], STD_RTOL)


def test_merge_tile_of_more_than_2_pow_32_elements(eng):
    """Maximum sizes: a tile of 65538 x 21846 x 3 = 4 295 229 444 elements (> 2^32; 4.3 GB per uint8 frame, 34 GB of float64 output).
    hm_merge splits it into row bands for the 32-bit-indexed streaming kernels: the result must equal, bit for bit, the same rows merged as
    explicit tiles by the caller, and the rows either side of the internal band boundary must match the oracle."""
    free, _ = torch.cuda.mem_get_info()
    if free < 100 * (1 << 30):
        pytest.skip("needs ~80 GB of free HBM")
    from camera_linearity_amd.synthetic import synthetic_icrf
    H, W, n = 65538, 21846, 2
    gen = torch.Generator(device="cuda").manual_seed(5)
    frames = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(n)]
    t = [1e-3, 4e-3]
    icrf, _ = synthetic_icrf()
    plan = eng.plan_merge(frames, t, icrf)
    assert plan.kernels.count("merge_u8_val3<N=2") == 2                      # two row bands, both through the streaming kernel
    plan.launch()
    whole = plan.outputs["val"]
    cut = 30000
    top = eng.merge([f[:cut] for f in frames], t, icrf, height=H, row0=0, rows=cut, buf_row0=0)["val"]
    assert torch.equal(top, whole[:cut])
    del top
    bot = eng.merge([f[cut:] for f in frames], t, icrf, height=H, row0=cut, rows=H - cut, buf_row0=cut)["val"]
    assert torch.equal(bot, whole[cut:])
    del bot
    r0 = 65530                                                               # the internal boundary is at row 65534
    ref = orc.merge([f[r0:].cpu().numpy() for f in frames], t, icrf)
    close(host(whole[r0:]), ref["val"], VAL_RTOL)


def test_correction_kernels_on_a_large_frame(eng):
    """Standalone hot-pixel filter, flat-field ROI mean, normalize_by_map and the Welford fold on a 1536 x 2048 x 3 frame: sizes at which
    every grid-stride / chunked loop of these kernels iterates (the small cases above finish in one pass), against the oracle."""
    rng = np.random.default_rng(29)
    h, w = 1536, 2048
    x = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    dark = rng.integers(0, 8, size=(h, w, 3)).astype(np.uint8)
    dark[rng.random((h, w, 3)) < 1e-3] = 200
    thr = 0.1
    for k in (3, 5):
        ref = orc.hot_pixel_filter(x.astype(np.float64), orc.unit_from_u8(dark), thr, k)
        out = eng.hot_pixel_filter(dev(x), dev(dark), thr, k)
        assert np.array_equal(host(out).astype(np.float64), ref)
    xs = rng.random((h, w, 3))
    refs = orc.hot_pixel_filter(xs, orc.unit_from_u8(dark), thr, 3)
    outs = eng.hot_pixel_filter(dev(xs), dev(orc.unit_from_u8(dark)), thr, 3)
    assert np.array_equal(host(outs), refs)
    # flat field: ROI means and normalize_by_map
    flat = np.clip(np.around(255 * (0.8 + 0.05 * rng.random((h, w, 3)))), 0, 255).astype(np.uint8)
    fval = orc.unit_from_u8(flat)
    fstd = 0.002 * (1 + rng.random((h, w, 3)))
    x0, x1, y0, y1 = eng.flat_roi_bounds(h, w, 0.2)
    m = host(eng.roi_mean(dev(flat), x0, x1, y0, y1))
    s = host(eng.roi_mean(dev(fstd), x0, x1, y0, y1))
    # NumPy's mean over axes (0, 1) of a (rows, cols, 3) slice accumulates ~1e5 terms per channel naively: the oracle is only good to
    # ~1e-12 here. Against long-double sums the kernel's tree reduction holds 1e-15 (uint8: the DN sum is exact in float64).
    close(m, orc.flat_roi_mean(fval, h, w, 0.2), 5e-12)
    close(s, orc.flat_roi_mean(fstd, h, w, 0.2), 5e-12)
    exact = lambda a: np.asarray(a[x0:x1, y0:y1, :].astype(np.longdouble).sum(axis=(0, 1)) / ((x1 - x0) * (y1 - y0)), dtype=np.float64)   # noqa: E731
    close(m, exact(flat.astype(np.longdouble) / np.longdouble(255)), 2e-15)
    close(s, exact(fstd), 2e-15)
    val, std = rng.random((h, w, 3)) + 0.1, 0.01 * (1 + rng.random((h, w, 3)))
    nv, ns = eng.normalize_by_map(dev(val), dev(std), dev(flat), dev(fstd), m, s)
    rv, rs = orc.normalize_by_map(val, std, fval, fstd, m, s)
    close(host(nv), rv, 1e-14)
    close(host(ns), rs, 1e-9)
    # Welford fold of 5 frames (one launch) and its finalisation
    frames = [rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(5)]
    mean = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    m2 = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
    c = eng.welford_update([dev(f) for f in frames], 0, mean, m2)
    om, om2, oc = orc.welford_state(frames)
    assert c == oc == 5
    assert np.array_equal(host(mean), om) and np.array_equal(host(m2), om2)


def test_headline_kernel_steady_state(eng):
    """The bench kernel itself at the bench size on NON-constant data: 7 x 4096 x 4096 x 3 val-only goes to merge_u8_val3<N=7,U=4,PF=1>
    and a wave needs more than 6.3 M elements (3072 workgroups x 4 waves x 512 elements) before it iterates and enters the RA / RB
    register ping-pong - smaller tests only ever run the prologue and the last-unit path. Held to: the generic kernel bit for bit
    (every element), two row tiles bit for bit (different unit -> wave assignment), a 64-row band of the oracle at three places."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
    n, H, W = 7, 4096, 4096
    frames, _, t = synthetic_stack_device(23, n, H, W, device="cuda")
    icrf, _ = synthetic_icrf()
    plan = eng.plan_merge(frames, t, icrf)
    assert plan.kernels == "merge_u8_val3<N=7,U=4,PF=1,MAP=3>"
    plan.launch()
    val = plan.outputs["val"]
    gen = eng.plan_merge(frames, t, icrf, variant=-1)
    assert gen.kernels == "merge_generic<f64in=0,std=0>"
    gen.launch()
    assert torch.equal(gen.outputs["val"], val)
    del gen
    cut = 2311                                     # 2311 * 12288 elements: not a multiple of the 512-element unit -> generic tail in the top tile
    top = eng.merge([f[:cut] for f in frames], t, icrf, height=H, row0=0, rows=cut, buf_row0=0)
    bot = eng.merge([f[cut:] for f in frames], t, icrf, height=H, row0=cut, rows=H - cut, buf_row0=cut)
    assert torch.equal(top["val"], val[:cut]) and torch.equal(bot["val"], val[cut:])
    del top, bot
    for r0 in (0, 2000, H - 64):
        ref = orc.merge([f[r0:r0 + 64].cpu().numpy() for f in frames], t, icrf)
        close(host(val[r0:r0 + 64]), ref["val"], VAL_RTOL)
    # uniform-random DNs (every LDS gather conflicts): same checks against the generic kernel
    frames, _, t = synthetic_stack_device(24, n, H, W, device="cuda", uniform_dn=True)
    a_ = eng.merge(frames, t, icrf)["val"]
    b_ = eng.merge(frames, t, icrf, variant=-1)["val"]
    assert torch.equal(a_, b_)


def test_config5_shape_eight_resident_stacks(eng):
    """BASELINE configs[4] on one GPU's share: 8 distinct resident 7 x 4096 x 4096 x 3 stacks merged back to back on one stream
    (what bench.py --workload cfg5 times), each output band-checked against the oracle and all eight different."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
    n, H, W = 7, 4096, 4096
    icrf, _ = synthetic_icrf()
    plans, stacks = [], []
    for k in range(8):
        frames, _, t = synthetic_stack_device(300 + k, n, H, W, device="cuda")
        plans.append(eng.plan_merge(frames, t, icrf))
        stacks.append((frames, t))
    for _ in range(2):
        for p in plans:
            p.launch()
    torch.cuda.synchronize()
    sums = set()
    for k, (p, (frames, t)) in enumerate(zip(plans, stacks)):
        assert p.kernels == "merge_u8_val3<N=7,U=4,PF=1,MAP=3>"
        r0 = 37 * k
        ref = orc.merge([f[r0:r0 + 32].cpu().numpy() for f in frames], t, icrf)
        close(host(p.outputs["val"][r0:r0 + 32]), ref["val"], VAL_RTOL)
        sums.add(int(p.outputs["val"].view(torch.int64).sum().item()))
    assert len(sums) == 8


def test_fast_division_fallback_branches(eng):
    """merge_u8_val3 divides with an unscaled reciprocal-Newton-Markstein sequence only while its table check proves every operand in
    range, and takes the IEEE division otherwise (a wave-uniform branch no other test reaches: their tables are positive and tame).
    Tables with negative, -0.0, denormal and huge entries, and exposures of 1e-120 / 1e120, must give the generic kernel's bits."""
    rng = np.random.default_rng(5)
    n, h, w = 7, 96, 130                                           # 37 440 elements: 73 units of 512 + a tail
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]
    fr = [dev(f) for f in frames]
    t_ok = list(1e-3 * 2.0 ** np.arange(n))
    base = np.linspace(0, 1, 256)[:, None] ** np.array([2.2, 2.0, 1.8])[None, :]
    cases = []
    neg = base.copy(); neg[5:40, 0] *= -1.0; neg[0, :] = -0.0                      # negative and -0.0 entries
    cases.append(("negative / -0.0 ICRF", neg, t_ok))
    den = base.copy(); den[1:30, 1] = 1e-310; den[200:, 2] = 1e-305                  # denormal entries: w * g leaves [2^-300, 2^300]
    cases.append(("denormal ICRF", den, t_ok))
    big = base.copy(); big[100:130, :] = 1e200
    cases.append(("huge ICRF", big, t_ok))
    cases.append(("tiny exposures", base, [1e-120 * 2.0 ** i for i in range(n)]))    # 1 / t above 2^300: the host-side check
    cases.append(("huge exposures", base, [1e120 * 2.0 ** i for i in range(n)]))
    cases.append(("in range (fast path)", base, t_ok))
    with np.errstate(all="ignore"):
        for name, icrf, t in cases:
            plan = eng.plan_merge(fr, t, icrf)
            assert plan.kernels.startswith("merge_u8_val3<N=7,U=4,PF=1,MAP=3>"), name
            plan.launch()
            gen = eng.merge(fr, t, icrf, variant=-1)["val"]
            a_, b_ = plan.outputs["val"], gen
            assert torch.equal(a_.view(torch.int64), b_.view(torch.int64)), name        # bit patterns, NaN / inf / -0 included
            ref = orc.merge(frames, t, icrf)["val"]
            fin = np.isfinite(ref) & (np.abs(ref) > 1e-290)
            np.testing.assert_allclose(host(a_)[fin], ref[fin], rtol=1e-12, err_msg=name)


def test_row_tile_set_single_gpu_assembly(eng):
    """RowTileSet with world_size = 1 and 8 tiles through launch() and assemble(): the side-stream D2H copies into the pinned image
    reproduce a whole-image merge bit for bit (val and std, hot-pixel halo rows included); assemble() hands out its own buffers
    (documented) and copy=True the caller's."""
    from camera_linearity_amd import parallel
    n, H, W, k = 5, 203, 64, 3                                     # 203 rows / 8 tiles: ragged tile heights
    frames, stds, t = orc.synthetic_stack(91, n, H, W, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(91)
    dark = (rng.random((H, W, 3)) < 0.01).astype(np.uint8) * 200
    fr, sd, dk = [dev(f) for f in frames], [dev(s) for s in stds], dev(dark)
    whole = eng.merge(fr, t, icrf, diff, sd, darks=[dk] * n, dark_min=[100] * n, median_k=k)
    tiles = parallel.RowTileSet(H, 8, rank=0, world_size=1, median_k=k)
    for tile in tiles.mine:
        b0, b1 = tiles.input_rows(tile)
        tiles.add_tile(tile, [f[b0:b1] for f in fr], t, icrf, diff, [s[b0:b1] for s in sd], darks=[dk[b0:b1]] * n, dark_min=[100] * n, median_k=k)
    assert tiles.mine == list(range(8))
    tiles.launch()
    val, std = tiles.assemble()
    assert val.is_pinned() and torch.equal(val, whole["val"].cpu()) and torch.equal(std, whole["std"].cpu())
    val2, std2 = tiles.assemble()
    assert val2.data_ptr() == val.data_ptr()                       # the same pinned buffers, reused
    val3, _ = tiles.assemble(copy=True)
    assert val3.data_ptr() != val.data_ptr() and torch.equal(val3, val)
    with pytest.raises(ValueError):
        parallel.gather_tiles({}, tiles.bounds, world_size=1)


@pytest.mark.parametrize("C", [1, 2, 3, 4])
@pytest.mark.parametrize("flat_u8", [True, False])
def test_normalize_by_map_channels_and_alignment(eng, C, flat_u8):
    """normalize_by_map for every channel count on an odd-sized frame (37 x 53 x C: the running channel counter of the element-pair
    loop wraps at every step), uint8 and float64 flat fields, with and without std, and a view that is not 16-byte aligned (scalar
    accesses). Against the oracle (measurand.py:585-604). (A burst-shaped version of this five-stream kernel was measured in round 3
    and dropped: 367.7 against 364.5 us on a 4096 x 4096 x 3 frame.)"""
    rng = np.random.default_rng(40 + C)
    h, w = 37, 53
    val, std = rng.random((h, w, C)) + 0.1, 0.01 * (1 + rng.random((h, w, C)))
    flat = rng.integers(150, 250, (h, w, C)).astype(np.uint8)
    fval = orc.unit_from_u8(flat) if flat_u8 else 0.6 + 0.3 * rng.random((h, w, C))
    fstd = 0.002 * (1 + rng.random((h, w, C)))
    m, s = list(0.7 + 0.1 * rng.random(C)), list(0.002 + 0.001 * rng.random(C))
    rv, rs = orc.normalize_by_map(val, std, fval, fstd, np.array(m), np.array(s))
    dflat = dev(flat) if flat_u8 else dev(fval)
    nv, ns = eng.normalize_by_map(dev(val), dev(std), dflat, dev(fstd), m, s)
    close(host(nv), rv, 1e-14)
    close(host(ns), rs, 1e-9)
    nv2, none = eng.normalize_by_map(dev(val), None, dflat, None, m)
    assert none is None and torch.equal(nv2, nv)
    big = torch.zeros(val.size + 1, dtype=torch.float64, device="cuda")
    view = big[1:].view(val.shape)
    view.copy_(dev(val))
    nv3, ns3 = eng.normalize_by_map(view, dev(std), dflat, dev(fstd), m, s)
    assert torch.equal(nv3, nv) and torch.equal(ns3, ns)


def test_hot_pixel_queue_randomised_against_the_workspace_free_pass(eng):
    """80 random configurations - frames 1..32, channels 1..4, uint8 and float64 frames, k = 3 / 5 / 7, maps from sparse to dense, shared
    and per-frame maps and thresholds, some frames without a map, std / flat / sum-of-weights on and off, whole images and row tiles with
    halo: the queue pass (scan -> patch per lane) and the workspace-free pass (one hot element per wave) must agree bit for bit."""
    rng = np.random.default_rng(2026)
    for case in range(80):
        n = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 20, 32]))
        C = int(rng.choice([1, 2, 3, 3, 3, 4]))
        h, w = int(rng.integers(3, 70)), int(rng.integers(3, 90))
        k = int(rng.choice([3, 3, 3, 5, 7]))
        f64 = bool(rng.random() < 0.3)
        with_std = bool(rng.random() < 0.6)
        use_flat = bool(rng.random() < 0.3)
        density = float(rng.choice([0.0, 1e-3, 0.02, 0.3, 1.0]))
        t = list(1e-3 * 2.0 ** np.arange(n) * (1 + 0.1 * rng.random(n)))
        frames = [rng.integers(0, 256, (h, w, C)).astype(np.uint8) for _ in range(n)]
        if f64:
            frames = [f.astype(np.float64) / 255 + 1e-4 * rng.random((h, w, C)) for f in frames]
        stds = [0.004 * (1 + rng.random((h, w, C))) for _ in range(n)] if with_std else None
        icrf = np.linspace(0, 1, 256)[:, None] ** (1.5 + 0.3 * np.arange(C))[None, :]
        diff = np.gradient(icrf, 2 / 255, axis=0)
        shared = dev((rng.random((h, w, C)) < density).astype(np.uint8) * 200)
        darks, mins = [], []
        for i in range(n):
            mode = rng.integers(0, 4)
            if mode == 0:
                darks.append(None); mins.append(256)
            elif mode == 1:
                darks.append(shared); mins.append(100)
            elif mode == 2:
                darks.append(shared); mins.append(int(rng.integers(1, 256)))        # the same map with another threshold: a distinct pair
            else:
                darks.append(dev((rng.random((h, w, C)) < density).astype(np.uint8) * int(rng.integers(1, 256)))); mins.append(int(rng.integers(1, 200)))
        if all(d is None for d in darks):
            darks[0], mins[0] = shared, 100
        kw = {}
        if use_flat:
            kw.update(flat=dev(rng.integers(120, 250, (h, w, C)).astype(np.uint8)), ff_mean=list(0.7 + 0.1 * rng.random(C)))
            if with_std:
                kw.update(flat_std=dev(0.002 * (1 + rng.random((h, w, C)))), ff_std_mean=list(0.002 + 0.001 * rng.random(C)))
        fr = [dev(f) for f in frames]
        sd = [dev(s) for s in stds] if with_std else None
        tile = {}
        if h >= 8 and rng.random() < 0.4:                                               # a row tile with its median halo
            r0, r1 = sorted(int(x) for x in rng.choice(np.arange(1, h), 2, replace=False))
            b0, b1 = max(0, r0 - k // 2), min(h, r1 + k // 2)
            fr = [f[b0:b1] for f in fr]
            sd = None if sd is None else [s[b0:b1] for s in sd]
            darks = [None if d is None else d[b0:b1] for d in darks]
            tile = dict(height=h, row0=r0, rows=r1 - r0, buf_row0=b0)
            for key in ("flat", "flat_std"):
                if key in kw:
                    kw[key] = kw[key][r0:r1]
        tag = f"case {case}: n={n} C={C} {h}x{w} k={k} f64={f64} std={with_std} flat={use_flat} density={density} tile={tile}"
        outs = []
        for queue in (True, False):
            outs.append(eng.merge(fr, t, icrf, diff if with_std else None, sd, darks=darks, dark_min=mins, median_k=k, want_sum_w=True,
                                  hot_queue=queue, **kw, **tile))
        for key in outs[0]:
            assert torch.equal(outs[0][key].view(torch.int64), outs[1][key].view(torch.int64)), tag + " " + key


@pytest.mark.parametrize("n", [3, 7, 15])
def test_val_only_flat_field_kernel(eng, n):
    """val-only merge with a uint8 flat field = merge_u8_val3's FLAT instantiation (the flat's DNs travel as one more byte stream,
    (val / F) * m in the epilogue): at 1536 x 2048 x 3 every wave iterates (register ping-pong); bit for bit the generic kernel, two row
    tiles bit for bit, a band of the oracle; a float64 flat field and a sum-of-weights output keep the older kernel."""
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark
    H, W = 1536, 2048
    frames, _, t = synthetic_stack_device(60 + n, n, H, W, device="cuda")
    icrf, _ = synthetic_icrf()
    flat, _, _ = synthetic_flat_dark(60 + n, H, W, device="cuda")
    flat[5, 7, 1] = 0                                               # F = 0: inf / NaN exactly where the generic kernel has them
    m = [0.78, 0.81, 0.8]
    plan = eng.plan_merge(frames, t, icrf, flat=flat, ff_mean=m)
    assert plan.kernels.startswith(f"merge_u8_val3<N={n},") and plan.kernels.endswith(",flat=1>")
    plan.launch()
    val = plan.outputs["val"]
    gen = eng.merge(frames, t, icrf, flat=flat, ff_mean=m, variant=-1)["val"]
    assert torch.equal(val.view(torch.int64), gen.view(torch.int64))
    cut = 701
    top = eng.merge([f[:cut] for f in frames], t, icrf, flat=flat[:cut], ff_mean=m, height=H, row0=0, rows=cut, buf_row0=0)["val"]
    bot = eng.merge([f[cut:] for f in frames], t, icrf, flat=flat[cut:], ff_mean=m, height=H, row0=cut, rows=H - cut, buf_row0=cut)["val"]
    assert torch.equal(top.view(torch.int64), val[:cut].view(torch.int64)) and torch.equal(bot.view(torch.int64), val[cut:].view(torch.int64))
    r0 = 640
    ref = orc.merge([f[r0:r0 + 48].cpu().numpy() for f in frames], t, icrf)["val"]
    with np.errstate(all="ignore"):
        want = (ref / orc.unit_from_u8(flat[r0:r0 + 48].cpu().numpy())) * np.array(m)
    close(host(val[r0:r0 + 48]), want, VAL_RTOL)
    assert "merge_u8_fast<" in eng.plan_merge(frames, t, icrf, flat=flat.double() / 255, ff_mean=m).kernels
    assert "merge_u8_fast<" in eng.plan_merge(frames, t, icrf, flat=flat, ff_mean=m, want_sum_w=True).kernels


@pytest.mark.parametrize("n", [2, 7, 15])
def test_monochrome_val_only_kernel(eng, n):
    """C = 1 (monochrome cameras), val-only, with and without a uint8 flat field = merge_u8_val3's one-column instantiations: a
    1536 x 2048 x 1 stack (every wave iterates), bit for bit the generic kernel and two row tiles, a band of the oracle."""
    rng = np.random.default_rng(70 + n)
    H, W = 1536, 2048
    t = list(1e-3 * 2.0 ** np.arange(n))
    g = torch.Generator(device="cuda").manual_seed(70 + n)
    rad = torch.rand((H, W, 1), generator=g, device="cuda", dtype=torch.float64) * 4
    frames = [torch.clamp(torch.round(rad * float(ti * 255 / (4 * t[n // 2]))), 0, 255).to(torch.uint8) for ti in t]
    icrf = np.linspace(0, 1, 256)[:, None] ** 2.2
    flat = torch.randint(150, 250, (H, W, 1), generator=g, device="cuda", dtype=torch.uint8)
    for kw in ({}, dict(flat=flat, ff_mean=[0.79])):
        plan = eng.plan_merge(frames, t, icrf, **kw)
        assert plan.kernels.startswith(f"merge_u8_val3<N={n},") and plan.kernels.endswith("C=1>"), plan.kernels
        plan.launch()
        val = plan.outputs["val"]
        gen = eng.merge(frames, t, icrf, variant=-1, **kw)["val"]
        assert torch.equal(val.view(torch.int64), gen.view(torch.int64))
        cut = 777
        tkw = (lambda a, b: dict(flat=flat[a:b], ff_mean=[0.79])) if kw else (lambda a, b: {})
        top = eng.merge([f[:cut] for f in frames], t, icrf, height=H, row0=0, rows=cut, buf_row0=0, **tkw(0, cut))["val"]
        bot = eng.merge([f[cut:] for f in frames], t, icrf, height=H, row0=cut, rows=H - cut, buf_row0=cut, **tkw(cut, H))["val"]
        assert torch.equal(top, val[:cut]) and torch.equal(bot, val[cut:])
        ref = orc.merge([f[100:148].cpu().numpy() for f in frames], t, icrf)["val"]
        if kw:
            ref = (ref / orc.unit_from_u8(flat[100:148].cpu().numpy())) * 0.79
        close(host(val[100:148]), ref, VAL_RTOL)
    # with std: merge_u8_fast_std's one-column instantiation, the same three checks
    diff = np.gradient(icrf, 2 / 255, axis=0)
    stds = [0.004 * (1 + torch.rand((H, W, 1), generator=g, device="cuda", dtype=torch.float64)) for _ in range(n)]
    plan = eng.plan_merge(frames, t, icrf, diff, stds)
    assert plan.kernels == f"merge_u8_fast_std<N={n},U=1,flat=0,sum_w=0,C=1>", plan.kernels
    plan.launch()
    gen = eng.merge(frames, t, icrf, diff, stds, variant=-1)
    assert torch.equal(plan.outputs["val"], gen["val"]) and torch.equal(plan.outputs["std"], gen["std"])
    ref = orc.merge([f[100:148].cpu().numpy() for f in frames], t, icrf, diff, stds=[s_[100:148].cpu().numpy() for s_ in stds])
    close(host(plan.outputs["val"][100:148]), ref["val"], VAL_RTOL)
    close(host(plan.outputs["std"][100:148]), ref["std"], STD_RTOL)
    # ... and with the flat field (uint8 and float64 flats): bit for bit the generic kernel
    fstd = 0.002 * (1 + torch.rand((H, W, 1), generator=g, device="cuda", dtype=torch.float64))
    for fl in (flat, flat.double() / 255):
        plan = eng.plan_merge(frames, t, icrf, diff, stds, flat=fl, flat_std=fstd, ff_mean=[0.79], ff_std_mean=[0.002])
        assert plan.kernels == f"merge_u8_fast_std<N={n},U=1,flat=1,sum_w=0,C=1>", plan.kernels
        plan.launch()
        gen = eng.merge(frames, t, icrf, diff, stds, flat=fl, flat_std=fstd, ff_mean=[0.79], ff_std_mean=[0.002], variant=-1)
        assert torch.equal(plan.outputs["val"], gen["val"]) and torch.equal(plan.outputs["std"], gen["std"])


# ------------------------------------------------------------------ stacks of more than HM_MAX_FRAMES frames (chunked path)
@pytest.mark.parametrize("chunk", [2, 3, 5])
@pytest.mark.parametrize("mode", ["val", "std", "full", "f64", "f64std", "sumw_only"])
def test_chunked_merge_bit_identical_to_one_launch(eng, chunk, mode):
    """variant = -chunk forces hm_merge_chunk.hip's path (running sums in memory between launches of `chunk` frames) on a 7-frame
    stack: every mode gives the bits of merge_generic (variant -1) - the chunking does not show in the result. Odd element count
    (two-elements-per-thread body + a one-element tail)."""
    n, h, w = 7, 37, 21
    frames, stds, t = orc.synthetic_stack(90 + chunk, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(chunk)
    f64 = mode.startswith("f64")
    fr = [dev(orc.unit_from_u8(f) + (rng.random(f.shape) - 0.5) * 1e-3) if f64 else dev(f) for f in frames]
    sd = [dev(s_) for s_ in stds] if mode in ("std", "full", "f64std") else None
    kw = {}
    if mode == "full":
        dark = (rng.random((h, w, 3)) < 0.03).astype(np.uint8) * 200
        dark2 = (rng.random((h, w, 3)) < 0.03).astype(np.uint8) * 220
        kw.update(darks=[None, dev(dark), dev(dark2), dev(dark), None, dev(dark2), dev(dark)], dark_min=[256, 100, 100, 100, 256, 100, 100], median_k=3,
                  flat=dev(rng.integers(180, 230, size=(h, w, 3)).astype(np.uint8)), flat_std=dev(np.full((h, w, 3), 0.002)),
                  ff_mean=[0.8, 0.81, 0.79], ff_std_mean=[0.002] * 3, want_sum_w=True)
    if mode == "sumw_only":
        kw.update(want_sum_w=True, want_val=False)
    a = eng.merge(fr, t, icrf, diff if sd else None, sd, variant=-chunk, **kw)
    b = eng.merge(fr, t, icrf, diff if sd else None, sd, variant=-1, **kw)
    assert set(a) == set(b) and len(a) >= 1
    for key in a:
        assert torch.equal(a[key], b[key]), (key, mode, chunk)
    # a view that starts at an odd byte / 8-byte-only aligned float64: the one-element-per-thread kernels
    if mode in ("val", "f64"):
        big = [torch.empty(h * w * 3 + 1, dtype=f_.dtype, device="cuda") for f_ in fr]
        fo = []
        for bt, f_ in zip(big, fr):
            bt[1:] = f_.reshape(-1)
            fo.append(bt[1:].view(h, w, 3))
        c = eng.merge(fo, t, icrf, variant=-chunk)
        assert torch.equal(c["val"], b["val"])


@pytest.mark.parametrize("n,with_std,f64", [(33, False, False), (40, True, False), (70, True, False), (45, False, True), (36, True, True)])
def test_merge_more_than_32_frames(eng, n, with_std, f64):
    """The reference's merge loop has no frame limit (modules/exposure_series.py:334,372): stacks of 33-70 frames against the oracle,
    with a dark map on some frames and a flat field. val rtol 1e-12 (uint8) / 1e-11 (float64 frames), std 1e-9."""
    h, w = 24, 40
    frames, stds, t = orc.synthetic_stack(200 + n, n, h, w, with_std=with_std)
    t = np.asarray(t) * 2.0 ** (-(n // 2))                      # keep the long exposures finite and the ratios of the recipe
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(n)
    dark = (rng.random((h, w, 3)) < 0.02).astype(np.uint8) * 210
    darks_h = [orc.unit_from_u8(dark) if i % 3 == 1 else None for i in range(n)]
    flat = rng.integers(180, 230, size=(h, w, 3)).astype(np.uint8)
    flat_std = np.full((h, w, 3), 0.002)
    m = orc.flat_roi_mean(orc.unit_from_u8(flat), h, w, 0.2)
    s_ = orc.flat_roi_mean(flat_std, h, w, 0.2)
    fh = [orc.unit_from_u8(f) for f in frames] if f64 else frames
    if with_std:
        ref = orc.merge(fh, t, icrf, diff, stds=stds, darks=darks_h, dark_threshold=0.5, median_k=3,
                        flat=orc.unit_from_u8(flat), flat_std=flat_std, ff_mean=m, ff_std_mean=s_)
    else:                                                        # val-only: measurand.py:602 alone
        ref = orc.merge(fh, t, icrf, None, darks=darks_h, dark_threshold=0.5, median_k=3)
        ref["val_ff"] = (ref["val"] / orc.unit_from_u8(flat)) * m
    dmin = eng.dark_min_dn(1.0, 0.5)
    out = eng.merge([dev(f) for f in fh], t, icrf, diff if with_std else None, [dev(x) for x in stds] if with_std else None,
                    darks=[dev(dark) if i % 3 == 1 else None for i in range(n)], dark_min=[dmin if i % 3 == 1 else 256 for i in range(n)], median_k=3,
                    flat=dev(flat), flat_std=dev(flat_std) if with_std else None, ff_mean=m, ff_std_mean=s_ if with_std else None)
    close(host(out["val"]), ref["val_ff"], F64_RTOL if f64 else VAL_RTOL)
    if with_std:
        close(host(out["std"]), ref["std_ff"], STD_RTOL)


# ------------------------------------------------------------------ the two builds of the C ABI against each other
@pytest.mark.parametrize("n,with_std,corr", [(3, False, False), (7, True, False), (7, True, True), (15, False, True), (20, True, True)])
def test_host_build_and_device_build_agree(eng, n, with_std, corr):
    """libhdrmerge_host.so (plain C++) and libhdrmerge.so (HIP) are two from-scratch builds of one ABI with one operation sequence per
    element: on uint8 stacks - tables from the host, IEEE division / sqrt, fma in frame order on both sides - they give the SAME BITS, with
    dark maps (hot-pixel medians) and a flat field too. (float64 frames evaluate exp() in two different math libraries: 1e-12.)"""
    from camera_linearity_amd.measurand import _HOST_ENGINE as heng
    h, w = 45, 38
    frames, stds, t = orc.synthetic_stack(300 + n, n, h, w, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    rng = np.random.default_rng(n)
    kw_h, kw_d = {}, {}
    if corr:
        dark = (rng.random((h, w, 3)) < 0.03).astype(np.uint8) * 210
        flat = rng.integers(170, 235, size=(h, w, 3)).astype(np.uint8)
        fstd = np.full((h, w, 3), 0.002)
        dk = [dark if i % 2 else None for i in range(n)]
        common = dict(dark_min=[90 if i % 2 else 256 for i in range(n)], median_k=3, ff_mean=[0.8, 0.79, 0.81])
        if with_std:
            common.update(ff_std_mean=[0.002] * 3)
        kw_h = dict(darks=[None if d is None else torch.from_numpy(d) for d in dk], flat=torch.from_numpy(flat), **common)
        kw_d = dict(darks=[None if d is None else dev(d) for d in dk], flat=dev(flat), **common)
        if with_std:
            kw_h.update(flat_std=torch.from_numpy(fstd))
            kw_d.update(flat_std=dev(fstd))
    sd_h = [torch.from_numpy(s_) for s_ in stds] if with_std else None
    sd_d = [dev(s_) for s_ in stds] if with_std else None
    host = heng.merge([torch.from_numpy(f) for f in frames], t, icrf, diff if with_std else None, sd_h, want_sum_w=True, **kw_h)
    devo = eng.merge([dev(f) for f in frames], t, icrf, diff if with_std else None, sd_d, want_sum_w=True, **kw_d)
    for key in host:
        assert np.array_equal(host[key].numpy(), devo[key].cpu().numpy()), key
    # float64 frames
    f64 = [orc.unit_from_u8(f) + (rng.random(f.shape) - 0.5) * 1e-3 for f in frames]
    host = heng.merge([torch.from_numpy(f) for f in f64], t, icrf, diff if with_std else None, sd_h)
    devo = eng.merge([dev(f) for f in f64], t, icrf, diff if with_std else None, sd_d)
    close(devo["val"].cpu().numpy(), host["val"].numpy(), 1e-12)
    if with_std:
        close(devo["std"].cpu().numpy(), host["std"].numpy(), STD_RTOL)


def test_plan_graph_replays_every_launch_bit_identically(eng):
    """engine.PlanGraph: plans of every launch shape (one streaming kernel; streaming + dark-map scan + patch; the chunked launches of a
    40-frame stack; a run-time-N stack with std and flat field) recorded into ONE hipGraph. A replay must reproduce the directly launched
    outputs bit for bit - also after the inputs were overwritten in place (the graph reads the tensors, not a snapshot of them)."""
    rng = np.random.default_rng(77)
    icrf, diff = orc.synthetic_icrf()
    plans, inputs = [], []
    # config-1-sized val-only stacks
    for seed in range(4):
        frames, _, t = orc.synthetic_stack(300 + seed, 3, 64, 48)
        fd = [dev(f) for f in frames]
        plans.append(eng.plan_merge(fd, t, icrf)); inputs.append(fd)
    # std + dark maps + flat field (three kernels per launch())
    frames, stds, t = orc.synthetic_stack(310, 7, 40, 36, with_std=True)
    d = rng.integers(0, 10, size=(40, 36, 3)).astype(np.uint8); d[rng.random(d.shape) < 0.02] = 210
    flat = rng.integers(150, 240, (40, 36, 3)).astype(np.uint8)
    fd = [dev(f) for f in frames]
    plans.append(eng.plan_merge(fd, t, icrf, diff, [dev(s) for s in stds], darks=[dev(d)] * 7, dark_min=[100] * 7, median_k=3,
                                flat=dev(flat), flat_std=dev(np.full(flat.shape, 0.002)), ff_mean=[0.8, 0.81, 0.79], ff_std_mean=[0.002] * 3))
    inputs.append(fd)
    # 40 frames: two chunked launches; 20 frames with std: the run-time-N kernel
    for n, with_std in ((40, False), (20, True)):
        frames, stds, t = orc.synthetic_stack(320 + n, n, 24, 32, with_std=with_std)
        fd = [dev(f) for f in frames]
        plans.append(eng.plan_merge(fd, t, icrf, diff if with_std else None, [dev(s) for s in stds] if with_std else None)); inputs.append(fd)
    graph = eng.PlanGraph(plans)
    wide = eng.PlanGraph(plans, lanes=3)               # the same launches on three parallel branches

    def direct():
        for p in plans:
            p.launch()
        torch.cuda.synchronize()
        return [{k: v.clone() for k, v in p.outputs.items()} for p in plans]

    def replayed(g):
        for p in plans:
            for v in p.outputs.values():
                v.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        return [{k: v.clone() for k, v in p.outputs.items()} for p in plans]

    for round_ in range(2):
        want = direct()
        for g in (graph, wide):
            for i, (a_, b_) in enumerate(zip(want, replayed(g))):
                assert a_.keys() == b_.keys()
                for key in a_:
                    assert torch.equal(a_[key], b_[key]), (round_, g.lanes, i, key)
                    assert not torch.isnan(b_[key]).any() or torch.isnan(a_[key]).any()
        for fd in inputs:                                   # new image data in the same tensors
            for f in fd:
                f.copy_(torch.flip(f, dims=(1,)))
    with pytest.raises(ValueError):
        eng.PlanGraph([])


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4tile"])
def test_full_size_device_output_equals_the_host_build(eng, cfg):
    """BASELINE.json's configurations AT FULL SIZE, every output element compared: the HIP kernels' result against the independent plain-C++
    build of the same ABI (libhdrmerge_host.so; itself pinned to the reference-generated goldens and the oracle in the CPU suite) -
    bit for bit. cfg2: 7 x 4096 x 4096 x 3 val-only (merge_u8_val3); cfg3: the same stack with float64 std frames, seven DISTINCT dark maps
    (k = 3 medians) and a flat field (merge_u8_fast_std + scan + patch); cfg4tile: rows 3072..4095 of a 15 x 8192 x 8192 x 3 image with
    the tile's halo rows. 50-100 million elements each; the host build takes a few seconds on the box's cores."""
    from camera_linearity_amd.measurand import _HOST_ENGINE as heng
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark
    icrf, diff = synthetic_icrf()
    cpu = lambda t: None if t is None else t.cpu()   # noqa: E731
    if cfg == "cfg4tile":
        n, H, W = 15, 1026, 8192                        # buffer rows 3071..4096 of the image: one halo row above and below the tile
        frames, _, t = synthetic_stack_device(11, n, H, W)
        kw = dict(height=8192, row0=3072, rows=1024, buf_row0=3071)
        d_out = eng.merge(frames, t, icrf, **kw)
        h_out = heng.merge([cpu(f) for f in frames], t, icrf, **kw)
    else:
        n, H, W = 7, 4096, 4096
        with_std = cfg == "cfg3"
        frames, stds, t = synthetic_stack_device(7, n, H, W, with_std=with_std)
        kw_d, kw_h = {}, {}
        if with_std:
            flat, flat_std, _ = synthetic_flat_dark(7, H, W)
            darks = [synthetic_flat_dark(20 + i, H, W, hot_density=1e-4)[2] for i in range(n)]
            x0, x1, y0, y1 = eng.flat_roi_bounds(H, W, 0.5)
            m = eng.roi_mean(flat, x0, x1, y0, y1).cpu().numpy()
            sm = eng.roi_mean(flat_std, x0, x1, y0, y1).cpu().numpy()
            common = dict(dark_min=[100] * n, median_k=3, ff_mean=m, ff_std_mean=sm)
            kw_d = dict(darks=darks, flat=flat, flat_std=flat_std, **common)
            kw_h = dict(darks=[cpu(d) for d in darks], flat=cpu(flat), flat_std=cpu(flat_std), **common)
        d_out = eng.merge(frames, t, icrf, diff if with_std else None, stds, **kw_d)
        h_out = heng.merge([cpu(f) for f in frames], t, icrf, diff if with_std else None, None if stds is None else [cpu(s) for s in stds], **kw_h)
    assert set(d_out) == set(h_out)
    for key in h_out:
        got = d_out[key].cpu()
        assert got.shape == h_out[key].shape and got.shape[0] == (1024 if cfg == "cfg4tile" else 4096)
        assert torch.equal(got, h_out[key]), (cfg, key, int((got != h_out[key]).sum()))
        assert bool(torch.isfinite(got).all())


def test_odd_width_tiles_keep_the_streaming_kernels(eng):
    """An image with an odd W * C (333 x 3) cut into row tiles with the k = 3 median's halo: with RowTileSet(row_elems=W * C) every tile
    starts an even number of elements into its buffer and is merged by the streaming kernel (the library's own dispatch says which); without
    it the odd-offset tiles fall to merge_generic. Either way the tiles are the whole image's rows, bit for bit."""
    from camera_linearity_amd import parallel
    rng = np.random.default_rng(9)
    n, H, W = 7, 600, 333
    frames, stds, t = orc.synthetic_stack(70, n, H, W, with_std=True)
    icrf, diff = orc.synthetic_icrf()
    dark = rng.integers(0, 10, (H, W, 3)).astype(np.uint8); dark[rng.random(dark.shape) < 0.01] = 220
    fd, sd, dd = [dev(f) for f in frames], [dev(s) for s in stds], dev(dark)
    whole = eng.merge(fd, t, icrf, diff, sd, darks=[dd] * n, dark_min=[100] * n, median_k=3)
    for row_elems, expect_generic in ((W * 3, False), (None, True)):
        tiles = parallel.RowTileSet(H, 5, median_k=3, row_elems=row_elems)
        saw_generic = False
        for tile in range(5):
            b0, b1 = tiles.input_rows(tile)
            # (.clone(): a buffer of the tile's own, as on another GPU - a view into the whole image would inherit ITS alignment)
            tiles.add_tile(tile, [f[b0:b1].clone() for f in fd], t, icrf, diff, [s[b0:b1].clone() for s in sd], darks=[dd[b0:b1].clone()] * n,
                           dark_min=[100] * n, median_k=3)
            kern = tiles.plans[tile].kernels
            saw_generic = saw_generic or kern.startswith("merge_generic")
            if row_elems is not None:
                assert kern.startswith("merge_u8_fast_std"), (tile, kern)
        assert saw_generic == expect_generic
        tiles.launch()
        torch.cuda.synchronize()
        for tile in range(5):
            r0, r1 = tiles.bounds[tile]
            for key in ("val", "std"):
                assert torch.equal(tiles.plans[tile].outputs[key], whole[key][r0:r1]), (row_elems, tile, key)
