"""CPU-only tests: the C-ABI library loads and exports every symbol include/hdrmerge.h declares, argument
validation that needs no device, host-side logic (settings, file-name grammar, dark selection, ROI bounds,
row tiles, weight tables) and constructor/type errors modelled on the reference's
tests/unit/test_measurand.py:120-167 and tests/unit/test_image_set.py:317-361."""
import ctypes as C
import os
import pathlib
import re

import numpy as np
import pytest

from oracle import hdr_oracle as orc

ROOT = pathlib.Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from camera_linearity_amd import _native as nat
    header = (ROOT / "include" / "hdrmerge.h").read_text()
    debug_header = (ROOT / "include" / "hdrmerge_debug.h").read_text()
    assert "hm_debug" not in header                      # the boundary header holds only what replaces a reference function
    declared = set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", header)) | set(re.findall(r"\b(hm_debug_[a-z0-9_]+)\s*\(", debug_header))
    declared -= {"hm_merge_args"}
    assert declared == set(nat.EXPORTED_SYMBOLS), declared ^ set(nat.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(nat.lib, name)
    assert nat.lib.hm_version() == nat.HM_ABI_VERSION == 2
    assert nat.strerror(0) == "ok" and "geometry" in nat.strerror(nat.HM_ESHAPE)
    assert "unknown" in nat.strerror(-99)


def test_merge_args_layout_matches_header():
    from camera_linearity_amd import _native as nat
    # offsets of include/hdrmerge.h's struct on LP64 (checked against a C compile in the build container)
    assert C.sizeof(nat.MergeArgs) == 296
    assert nat.MergeArgs.frames_u8.offset == 64 and nat.MergeArgs.ff_mean.offset == 176
    assert nat.MergeArgs.out_sum_w.offset == 256 and nat.MergeArgs.hot_workspace.offset == 264
    assert nat.MergeArgs.frames_workspace.offset == 280


def test_argument_validation_without_device():
    """Validation happens before any HIP call, so it can be exercised on a box without a GPU."""
    from camera_linearity_amd import _native as nat
    assert nat.lib.hm_merge(None, None) == nat.HM_EINVAL
    a = nat.MergeArgs()
    a.struct_size = 1
    assert nat.lib.hm_merge(C.byref(a), None) == nat.HM_EINVAL            # wrong struct size
    a.struct_size = C.sizeof(nat.MergeArgs)
    a.n_frames, a.channels, a.height, a.width, a.rows, a.buf_rows = 2, 5, 4, 4, 4, 4
    assert nat.lib.hm_merge(C.byref(a), None) == nat.HM_EUNSUPPORTED      # > HM_MAX_CHANNELS
    a.channels = 3
    assert nat.lib.hm_merge(C.byref(a), None) == nat.HM_EINVAL            # no frames given
    assert nat.lib.hm_merge_algorithmic_bytes(C.byref(a)) == 2 * 48       # 2 uint8 frames, no outputs requested
    assert nat.lib.hm_linearize_u8(None, None, None, None, None, None, 0, 3, 3, None) == nat.HM_EINVAL
    assert nat.lib.hm_binary_op(9, None, None, None, None, None, None, 1, None, None, None, None) == nat.HM_EINVAL
    assert nat.lib.hm_roi_mean_workspace_bytes() >= 8 * 4
    # the 8(f) entry points validate before any device work too
    one = (C.c_void_p * 1)(None)
    assert nat.lib.hm_welford_update(one, 40, 0, None, None, None, 12, 3, None) == nat.HM_EINVAL          # > HM_MAX_FRAMES
    assert nat.lib.hm_welford_update(one, 1, 0, None, None, None, 12, 5, None) == nat.HM_ESHAPE           # C > 4
    assert nat.lib.hm_welford_update(one, 1, 0, None, None, None, 10, 3, None) == nat.HM_ESHAPE           # n_elems % C != 0
    assert nat.lib.hm_welford_update(one, 0, 0, None, None, None, 12, 3, None) == nat.HM_OK               # nothing to fold
    assert nat.lib.hm_welford_update(one, 1, 0, None, None, None, 12, 3, None) == nat.HM_EINVAL           # null state / frame
    assert nat.lib.hm_welford_finalize(None, None, 0, None, None, 12, None) == nat.HM_EINVAL              # count < 1
    assert nat.lib.hm_welford_algorithmic_bytes(16, 1, 100) == 100 * (16 + 32)
    t3 = (C.c_double * 3)(1.0, 2.0, 4.0)
    assert nat.lib.hm_linearity_energy(None, None, t3, None, None, 4, 5, 250, 1, 16, 1, None, None, None, None) == nat.HM_ESHAPE   # one frame
    assert nat.lib.hm_linearity_energy(None, None, t3, None, None, 4, 5, 300, 1, 16, 3, None, None, None, None) == nat.HM_EINVAL  # limit > 255
    assert nat.lib.hm_linearity_energy(None, None, t3, None, None, 0, 5, 250, 1, 16, 3, None, None, None, None) == nat.HM_OK      # no candidates
    assert nat.lib.hm_linearity_energy(None, None, t3, None, None, 4, 5, 250, 1, 16, 3, None, None, None, None) == nat.HM_EINVAL  # null buffers
    assert nat.lib.hm_linearity_energy_workspace_bytes(1 << 20, 7, 75) == 75 * 21 * 64 * 16
    assert nat.lib.hm_tiff_lzw_decode(None, 0, None, 0) == nat.HM_EINVAL
    with pytest.raises(ValueError):
        nat.check(nat.HM_ESHAPE)
    with pytest.raises(NotImplementedError):
        nat.check(nat.HM_EUNSUPPORTED)
    with pytest.raises(nat.HdrMergeError):
        nat.check(nat.HM_ELAUNCH)


def test_host_weight_lut_helper_close_to_numpy():
    from camera_linearity_amd import _native as nat
    w = (C.c_double * 256)()
    dw = (C.c_double * 256)()
    assert nat.lib.hm_gaussian_weight_lut_host(w, dw) == 0
    rw, rdw = orc.gaussian_weight_lut()
    np.testing.assert_allclose(np.array(w), rw, rtol=4e-16)
    np.testing.assert_allclose(np.array(dw), rdw, rtol=4e-16)


def test_engine_weight_tables_bit_identical_to_reference_values(golden):
    from camera_linearity_amd import engine
    w, dw = engine.weight_luts_host()
    g = golden("merge_ramp")
    assert np.array_equal(w, g["w_lut"]) or np.allclose(w, g["w_lut"], rtol=2e-16, atol=0)
    np.testing.assert_allclose(dw, g["dw_lut"], rtol=2e-16)


def test_dark_min_dn_equivalent_to_value_threshold():
    from camera_linearity_amd import engine
    for scale, thr in ((1.0, 0.05), (0.4, 0.03), (0.8, 0.012), (1.0, 2.0), (1.0, -1.0)):
        v = orc.unit_from_u8(np.arange(256, dtype=np.uint8))
        v = v if scale == 1.0 else np.array([scale]) * v
        hot = v > thr
        m = engine.dark_min_dn(scale, thr)
        assert np.array_equal(np.arange(256) >= m, hot)


def test_roi_bounds_and_row_tiles():
    from camera_linearity_amd import engine, parallel
    assert engine.flat_roi_bounds(32, 48, 0.2) == orc.flat_roi_bounds(32, 48, 0.2) == (12, 18, 18, 27)
    for H in (0, 1, 7, 4096, 8191):
        for G in (1, 2, 3, 8):
            b = parallel.row_tile_bounds(H, G)
            assert b[0][0] == 0 and b[-1][1] == H and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(r1 - r0 for r0, r1 in b) - min(r1 - r0 for r0, r1 in b) <= 1
    assert parallel.halo_bounds(0, 10, 40, 3) == (0, 11) and parallel.halo_bounds(10, 40, 40, 5) == (8, 40)
    assert parallel.halo_bounds(10, 20, 40, 0) == (10, 20)
    assert parallel.stacks_for_rank(10, 1, 4) == [1, 5, 9]


def test_file_name_features():
    """The table of the reference's tests/unit/test_image_set.py:317-327."""
    from camera_linearity_amd.image_set import _features_from_file_name
    P = pathlib.Path
    assert _features_from_file_name(P("5ms BF sample_1 50x.tif")) == {
        "illumination": "BF", "magnification": "50x", "exposure": 0.005, "subject": "sample_1"}
    f = _features_from_file_name(P("/data/x/df 0.5ms 10X rock.tif"))
    assert f == {"illumination": "df", "magnification": "10X", "exposure": 0.0005, "subject": "rock"}
    assert _features_from_file_name(P("flat.tif")) == {"illumination": "", "magnification": "", "exposure": 0.0, "subject": "flat"}


def test_measurand_constructor_errors_and_properties():
    """modules/measurand.py:55-56,68-69,84,698-710 / tests/unit/test_measurand.py:120-167."""
    import torch
    from camera_linearity_amd.measurand import HipMeasurand, is_broadcastable
    from camera_linearity_amd.measurand_factory import Measurand
    with pytest.raises(TypeError):
        HipMeasurand("invalid")
    with pytest.raises(TypeError):
        HipMeasurand(np.ones(3), "invalid")
    with pytest.raises(ValueError):
        HipMeasurand(torch.ones(3, dtype=torch.float64), torch.ones(4, dtype=torch.float64))
    m = HipMeasurand(2.0, 0.5)
    assert tuple(m.val.shape) == (1,) and m.val.dtype == torch.float64 and float(m.std[0]) == 0.5
    with pytest.raises(AttributeError):
        m.channels = None
    with pytest.raises(TypeError):
        m.val = [1, 2, 3]
    with pytest.raises(TypeError):
        m.std = "x"
    m.val = np.zeros((4, 5, 3))
    assert list(m.channels.cpu().numpy()) == [0, 1, 2]                   # deviation B: arange(shape[-1])
    assert HipMeasurand(torch.zeros((2, 2, 4), dtype=torch.float64)).channels.numel() == 4
    hm = Measurand(np.zeros(3), use_cupy=False)                          # the host backend in the reference's NumPy slot
    assert hm.backend == "numpy" and isinstance(hm.val, np.ndarray) and type(hm).__name__ == "HostMeasurand"
    with pytest.raises(TypeError, match="Invalid other type"):
        m._normalize_input(hm)                                           # backends do not mix (sibling classes in the reference)
    assert Measurand().val is None and Measurand().backend == "hip"
    assert is_broadcastable((4, 1, 3), (5, 3)) and not is_broadcastable((4, 2), (3,))
    with pytest.raises(ValueError):
        is_broadcastable((), (1,))
    with pytest.raises(TypeError):
        m._normalize_input("abc")


def test_image_set_backend_checks_and_dark_selection():
    from camera_linearity_amd.image_set import ImageSet
    from camera_linearity_amd.measurand import HipMeasurand

    class Fake:
        backend = "cupy"
    with pytest.raises(ValueError):
        ImageSet(measurand=Fake())
    assert ImageSet().use_cupy is False and ImageSet().measurand.backend == "numpy"      # image_set.py:29: the reference's default
    s = ImageSet(use_cupy=True, value=np.zeros((2, 2, 3), np.uint8), features={"exposure": 0.04, "illumination": "bf", "magnification": "5x", "subject": "a"})
    with pytest.raises(AttributeError):
        s.use_cupy = False
    with pytest.raises(ValueError):
        s.measurand = Fake()
    assert s.measurand.dn is not None and s.measurand.dn.dtype.is_floating_point is False
    mk = lambda e: ImageSet(features={"exposure": e, "illumination": "bf", "magnification": "5x", "subject": "dark"},   # noqa: E731
                            measurand=HipMeasurand())
    darks = [mk(0.010), mk(0.050), mk(0.100)]
    for t_exp in (0.020, 0.040, 0.050, 0.080, 0.2, 0.005):
        s.features["exposure"] = t_exp
        d, sc = s.select_dark_field(darks, 0.03)
        j, osc = orc.select_dark(t_exp, [0.010, 0.050, 0.100], 0.03)
        assert (d is None and j < 0) or d is darks[j]
        assert sc == pytest.approx(osc, rel=1e-15)
    other = ImageSet(features=dict(s.features, exposure=1.0), measurand=HipMeasurand())
    assert s.is_exposure_match(other) and not s.is_exposure_match(mk(1.0))


def test_settings_configure():
    from camera_linearity_amd import settings
    old = settings.DARK_THRESHOLD
    settings.configure(DARK_THRESHOLD=0.012)
    assert settings.DARK_THRESHOLD == 0.012
    settings.configure(DARK_THRESHOLD=old)
    with pytest.raises(KeyError):
        settings.configure(NOPE=1)
    with pytest.raises(NotImplementedError):
        settings.configure(BITS=1024)


def test_oracle_is_test_infrastructure_only():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import the oracle: the product package,
    the tools and the examples must not (a product path that routes through the oracle would void every parity claim)."""
    import pathlib
    import re
    root = pathlib.Path(__file__).resolve().parent.parent
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    offenders = []
    for sub in ("camera_linearity_amd", "tools", "examples"):
        for f in (root / sub).rglob("*.py"):
            if pat.search(f.read_text()):
                offenders.append(str(f.relative_to(root)))
    assert offenders == []
    # bench.py and __graft_entry__.py import it only inside the baseline / smoke functions, never at module level
    for name in ("bench.py", "__graft_entry__.py"):
        src = (root / name).read_text()
        assert not re.search(r"^(from\s+oracle\b|import\s+oracle\b)", src, re.M), name


# ---------------------------------------------------------------------------------------------- dispatch (no GPU needed)
def _describe(n=7, C=3, H=64, W=64, std=False, flat=False, sumw=False, f64=False, variant=0, align=0, darks=False, hot_ws=False,
              frames_ws=False, struct_size=None):
    """hm_merge_describe runs hm_merge's own dispatch with launching switched off: pointers only need to look aligned."""
    import ctypes as C_
    from camera_linearity_amd import _native as nat
    a = nat.MergeArgs()
    a.struct_size = C_.sizeof(nat.MergeArgs) if struct_size is None else struct_size
    a.n_frames, a.channels, a.variant = n, C, variant
    a.height = a.rows = a.buf_rows = H
    a.width = W
    base = 1 << 20
    fr = (C_.c_void_p * n)(*[base * (i + 1) + align for i in range(n)])
    if f64:
        a.frames_f64 = C_.cast(fr, C_.POINTER(C_.c_void_p))
    else:
        a.frames_u8 = C_.cast(fr, C_.POINTER(C_.c_void_p))
    ex = (C_.c_double * n)(*[1e-3 * 2 ** i for i in range(n)])
    a.exposures = C_.cast(ex, C_.POINTER(C_.c_double))
    a.icrf, a.w_lut, a.out_val = base * 40, base * 41, base * 42
    keep = [fr, ex]
    if std:
        sd = (C_.c_void_p * n)(*[base * (50 + i) for i in range(n)])
        a.stds = C_.cast(sd, C_.POINTER(C_.c_void_p))
        a.icrf_diff, a.dw_lut, a.out_std = base * 43, base * 44, base * 45
        keep.append(sd)
    if flat:
        a.flat_u8 = base * 46
        a.flat_std = base * 47
    if sumw:
        a.out_sum_w = base * 48
    if darks:
        dk = (C_.c_void_p * n)(*[base * 90 for _ in range(n)])
        dm = (C_.c_int32 * n)(*[13] * n)
        a.darks_u8 = C_.cast(dk, C_.POINTER(C_.c_void_p))
        a.dark_min_dn = C_.cast(dm, C_.POINTER(C_.c_int32))
        a.median_k = 3
        keep += [dk, dm]
        if hot_ws:
            a.hot_workspace, a.hot_workspace_bytes = base * 91, nat.lib.hm_merge_hot_workspace_bytes(H * W * C)
    if frames_ws:
        a.frames_workspace, a.frames_workspace_bytes = base * 92, 8 * H * W * C
    buf = C_.create_string_buffer(512)
    rc = nat.lib.hm_merge_describe(C_.byref(a), buf, 512)
    return rc, buf.value.decode()


def test_merge_dispatch_table():
    """Which kernel hm_merge launches for which arguments (the library's own dispatch, dry): the val-only bench shape goes to
    merge_u8_val3 with the per-N configuration, std / flat / sum-of-weights to merge_u8_fast(_std), N > 16 and C != 3 to the
    run-time-N kernels, float64 frames to merge_f64_*, unaligned frames to merge_generic, dark maps add the fix-up pass."""
    from camera_linearity_amd import _native as nat
    assert _describe(7) == (0, "merge_u8_val3<N=7,U=4,PF=1,MAP=3>")
    assert _describe(8)[1] == "merge_u8_val3<N=8,U=3,PF=1,MAP=3>"             # (N = 8: three sub-units per wave, profiles/r04x_sweep8.log)
    assert _describe(15)[1] == "merge_u8_val3<N=15,U=3,PF=0,MAP=0>"
    assert _describe(7, H=4, W=8)[1] == "merge_generic<f64in=0,std=0>"                      # 96 elements: less than one group
    assert _describe(7, H=5, W=64)[1] == "merge_u8_val3<N=7,U=4,PF=1,MAP=3> + merge_generic<f64in=0,std=0>"   # 960 = 1 unit of 512 + tail
    assert _describe(7, std=True)[1] == "merge_u8_fast_std<N=7,U=1,flat=0,sum_w=0>"
    assert _describe(7, std=True, flat=True, darks=True)[1] == "merge_u8_fast_std<N=7,U=1,flat=1,sum_w=0> + merge_fixup_hot<f64in=0,std=1>"
    # with a queue workspace the hot-pixel pass is scan -> balanced patch (-> the old pass, gated on the queue's overflow flag)
    assert _describe(7, std=True, darks=True, hot_ws=True)[1] == ("merge_u8_fast_std<N=7,U=1,flat=0,sum_w=0> + merge_scan_hot + merge_patch_hot<f64in=0,std=1>"
                                                                 "")
    # workspace = 16 bytes of counters + the piece table (8 bytes per 65 536 elements + 1025 spare slots) + 4 bytes per queued element
    E = 4096 * 4096 * 3
    slots = E // 65536 + 1 + 1024
    assert nat.lib.hm_merge_hot_workspace_bytes(E) == 16 + 8 * slots + E                        # queue: a quarter of the elements
    assert nat.lib.hm_merge_hot_workspace_min_bytes(E) == 16 + 8 * slots + 4                    # room for one entry
    assert nat.lib.hm_merge_hot_workspace_bytes(100) == 16 + 8 * 1025 + 400 and nat.lib.hm_merge_hot_workspace_min_bytes(0) == 16 + 8 * 1025 + 4
    assert _describe(7, sumw=True)[1] == "merge_u8_fast<N=7,U=2,flat=0,sum_w=1>"
    assert _describe(7, flat=True)[1] == "merge_u8_val3<N=7,U=2,PF=1,MAP=3,flat=1>"             # val-only with a uint8 flat field: the bench kernel's FLAT instantiation
    assert _describe(15, flat=True)[1] == "merge_u8_val3<N=15,U=2,PF=1,MAP=0,flat=1>"
    assert _describe(7, flat=True, sumw=True)[1] == "merge_u8_fast<N=7,U=2,flat=1,sum_w=1>"
    assert _describe(17)[1] == "merge_u8_val3<N=17,U=3,PF=0,MAP=0>" and _describe(20, std=True)[1] == "merge_u8_fast_std<N=20,U=1,flat=0,sum_w=0>"   # templated up to N = 20
    assert _describe(21)[1] == "merge_u8_loop<C=3,flat=0,sum_w=0>(N=21)" and _describe(32, std=True)[1] == "merge_u8_loop_std<C=3,flat=0,sum_w=0>(N=32)"
    assert _describe(7, C=1)[1] == "merge_u8_val3<N=7,U=4,PF=1,MAP=3,C=1>"                  # monochrome val-only: the bench kernel with one table column
    assert _describe(7, C=1, flat=True)[1] == "merge_u8_val3<N=7,U=2,PF=1,MAP=3,flat=1,C=1>"
    assert _describe(7, C=1, std=True)[1] == "merge_u8_fast_std<N=7,U=1,flat=0,sum_w=0,C=1>"
    assert _describe(7, C=1, std=True, flat=True)[1] == "merge_u8_fast_std<N=7,U=1,flat=1,sum_w=0,C=1>"
    assert _describe(7, C=1, std=True, sumw=True)[1] == "merge_u8_loop_std<C=1,flat=0,sum_w=1>(N=7)"
    assert _describe(17, C=1)[1] == "merge_u8_loop<C=1,flat=0,sum_w=0>(N=17)" and _describe(7, C=2)[1] == "merge_u8_loop<C=2,flat=0,sum_w=0>(N=7)"
    assert _describe(7, f64=True, std=True)[1] == "merge_f64_std<C=3,flat=0,sum_w=0>(N=7)"
    assert _describe(7, align=1)[1] == "merge_generic<f64in=0,std=0>"
    assert _describe(7, variant=-1)[1] == "merge_generic<f64in=0,std=0>"
    assert _describe(7, variant=1120)[1] == "merge_u8_fast<N=7,U=2,flat=0,sum_w=0>"        # round 1's kernel stays reachable for A/B runs
    # more than HM_MAX_FRAMES frames: 32 per launch with the running sums in memory - the sum of weights in out_sum_w or the workspace
    assert _describe(33)[0] == nat.HM_EUNSUPPORTED                                           # neither given
    assert _describe(33, frames_ws=True) == (0, "merge_chunk<f64in=0,val,hot=0,vec=2>(N=33 in chunks of 32)")
    assert _describe(70, std=True, sumw=True, darks=True)[1] == "merge_chunk<f64in=0,S + std,hot=1,vec=2>(N=70 in chunks of 32)"
    assert _describe(40, f64=True, frames_ws=True, align=8)[1] == "merge_chunk<f64in=1,val,hot=0,vec=1>(N=40 in chunks of 32)"
    assert _describe(7, variant=-3, frames_ws=True)[1] == "merge_chunk<f64in=0,val,hot=0,vec=2>(N=7 in chunks of 3)"
    assert nat.lib.hm_merge_frames_workspace_bytes(32, 1000, 0) == 0 and nat.lib.hm_merge_frames_workspace_bytes(33, 1000, 0) == 8000
    assert nat.lib.hm_merge_frames_workspace_bytes(33, 1000, 1) == 0
    # callers built against the two ABI-1 layouts of hm_merge_args (264 bytes: no hot-pixel workspace; 280: with it) still work
    assert _describe(7, struct_size=264) == (0, "merge_u8_val3<N=7,U=4,PF=1,MAP=3>")
    assert _describe(7, std=True, darks=True, hot_ws=True, struct_size=264)[1].endswith("merge_fixup_hot<f64in=0,std=1>")   # the workspace fields are past the old struct's end
    assert _describe(7, std=True, darks=True, hot_ws=True, struct_size=280)[1].endswith("merge_patch_hot<f64in=0,std=1>")
    assert _describe(7, struct_size=272)[0] == nat.HM_EINVAL


def test_probe_and_tuning_variants_are_rejected_by_the_default_library():
    """The shipped library is built without the tuning matrix: the table-free traffic probe (variant 5120: wrong results by design)
    and the A/B variants of merge_u8_val3 / merge_u8_priv must come back as HM_EINVAL, not run."""
    from camera_linearity_amd import _native as nat
    for v in (5120, 5020, 7300, 7211, 8200, 8410, 999999):
        assert _describe(7, variant=v)[0] == nat.HM_EINVAL, v


def test_merge_splits_huge_tiles_into_row_bands():
    """A tile of 2^32 elements or more (the streaming kernels index with 32 bits) is dispatched as row bands, each through the
    streaming kernel - not handed to merge_generic as in round 1."""
    rc, names = _describe(7, H=65538, W=21846)                     # 65538 * 65538 = 4 295 229 444 elements > 2^32
    assert rc == 0
    parts = names.split(" + ")
    assert parts[0] == "merge_u8_val3<N=7,U=4,PF=1,MAP=3>" and parts.count("merge_u8_val3<N=7,U=4,PF=1,MAP=3>") == 2, names
    assert all(p.startswith("merge_u8_val3") or p.startswith("merge_generic") for p in parts)
    rc, names = _describe(7, H=65538, W=21846, std=True, darks=True)
    assert rc == 0 and names.count("merge_u8_fast_std") == 2 and names.count("merge_fixup_hot") == 2


# ---------------------------------------------------------------------------------------------- bench.py launcher (no GPU needed)
def test_bench_self_launch_spawns_ranks_before_touching_torch():
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) must start the ranks as a CHILD
    `python -m torch.distributed.run ... bench.py <same arguments>` and return its exit code - before torch (and with it any HIP
    call) is imported into the parent: a process that has initialised the GPU must never exec or fork workers on the pool's boxes."""
    import pathlib
    import subprocess
    import sys
    import textwrap
    root = pathlib.Path(__file__).resolve().parent.parent
    code = textwrap.dedent(f"""
        import json, os, sys
        sys.path.insert(0, {str(root)!r})
        os.environ.pop("WORLD_SIZE", None)
        import bench
        assert "torch" not in sys.modules and "numpy" not in sys.modules, "bench.py imports torch at module level"
        seen = {{}}
        class R:
            returncode = 42
        def fake(cmd, env=None, **kw):
            seen["cmd"], seen["torch_loaded"] = cmd, "torch" in sys.modules
            seen["ipc"] = (env or {{}}).get("HSA_ENABLE_IPC_MODE_LEGACY")
            return R()
        bench.subprocess.run = fake
        sys.argv = ["bench.py", "--gpus", "4", "--workload", "cfg4", "--steps", "3"]
        try:
            bench.main()
        except SystemExit as e:
            seen["rc"] = e.code
        seen["torch_after"] = "torch" in sys.modules
        print(json.dumps(seen))
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    seen = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = seen["cmd"]
    assert seen["rc"] == 42 and seen["torch_loaded"] is False and seen["torch_after"] is False
    assert seen["ipc"] == "0"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = [i for i, c in enumerate(cmd) if c.endswith("bench.py")][0]
    assert cmd[k + 1:] == ["--gpus", "4", "--workload", "cfg4", "--steps", "3"]


def test_bench_does_not_self_launch_under_a_launcher():
    """With WORLD_SIZE set (torch.distributed.run started us) main() goes on to initialise its rank instead of spawning again;
    a mismatch between --gpus and the launcher's world size is an error, not a silent run."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_bench_counter_records_go_stale_with_the_kernel_source(tmp_path, monkeypatch):
    """bench.py prints a PMC record (roofline.traffic, roofline_valu) only while the hash stored with it is the hash of the kernel's
    source files in this tree; a record collected on other sources reads as traffic None + traffic_stale True."""
    import importlib
    import json
    bench = importlib.import_module("bench")
    h = bench.kernel_source_hash("cfg2")
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == bench.kernel_source_hash("cfg3flat") and h != bench.kernel_source_hash("linearitystd")
    assert bench.kernel_source_hash("welford") != bench.kernel_source_hash("energy")
    rec = tmp_path / "rec.json"
    monkeypatch.setattr(bench, "COUNTER_RECORDS", str(rec))
    assert bench.measured_traffic("cfg2") == (None, None, False)                   # no record at all
    json.dump({"cfg2": {"hbm_bytes_per_launch": 755e6, "kernel": "k", "commit": "abc", "source_hash": h},
               "cfg3": {"hbm_bytes_per_launch": 4.9e9, "kernel": "k", "commit": "abc", "source_hash": "0" * 16},
               "linearitystd": {"valu_wave_instructions_per_launch": 1.0e9, "kernel": "k", "commit": "abc", "source_hash": bench.kernel_source_hash("linearity")}},
              open(rec, "w"))
    t, src, stale = bench.measured_traffic("cfg2")
    assert t == 755e6 and "abc" in src and stale is False
    assert bench.measured_traffic("cfg3") == (None, None, True)                    # collected on other sources
    blk = bench.valu_roofline("linearitystd", 2000.0, ("instructions_per_element_pair", 64e6))
    assert blk["frac"] == round(1.0e9 / 2000.0 / 1e3 / bench.VALU_PEAK_GINSTR, 4) and blk["instructions_per_element_pair"] == 1000.0
    assert bench.valu_roofline("welford", 100.0)["frac"] is None


def test_hm_merge_survives_random_argument_structs():
    """2 000 random hm_merge_args - wrong struct sizes, negative / huge / inconsistent geometry, NULL and misaligned pointers, too many or no
    frames, both frame kinds at once, bad median sizes, NaN / zero / negative exposures - through hm_merge, hm_merge_describe and
    hm_merge_algorithmic_bytes of the HIP library on a machine WITHOUT a GPU: every call returns (HM_OK for calls with nothing to do, a
    negative HM_E* code otherwise), none crashes. Device pointers are fake addresses: validation must never dereference them."""
    import random
    from camera_linearity_amd import _native as nat
    lib = nat.hip_lib
    rng = random.Random(1)
    seen = {}

    def fake():
        return 0x7f0000000000 + rng.randrange(0, 1 << 20) * 16 + rng.choice([0, 0, 0, 1, 3, 8])
    for _ in range(2000):
        a = nat.MergeArgs()
        a.struct_size = rng.choice([C.sizeof(nat.MergeArgs)] * 6 + [264, 280, 0, 100, 1024])
        n = rng.choice([1, 2, 7, 15, 16, 17, 32, 33, 40, 0, -1, 1000])
        a.n_frames, a.channels = n, rng.choice([3, 3, 3, 1, 2, 4, 0, 5, -2])
        a.variant = rng.choice([0, 0, 0, -1, -2, -5, 1120, 7300, 99999, -100])
        H, W = rng.choice([0, 1, 5, 64, 4096, -3, 70000]), rng.choice([0, 1, 7, 64, 4096, -1, 1 << 20])
        a.height, a.width = H, W
        a.row0 = rng.choice([0, 0, 1, H // 2 if H > 0 else 0, -1, H + 5])
        a.rows = rng.choice([H, max(H, 0) // 2, 0, -1, H + 1])
        a.buf_row0 = rng.choice([0, 0, a.row0, a.row0 - 1, -2])
        a.buf_rows = rng.choice([H, a.rows, a.rows + 2, 0, -1])
        nn = max(1, min(abs(n), 64))
        ptrs = (C.c_void_p * nn)(*[fake() if rng.random() < 0.9 else None for _ in range(nn)])
        kind = rng.choice(["u8", "f64", "none", "both"])
        if kind in ("u8", "both"):
            a.frames_u8 = C.cast(ptrs, C.POINTER(C.c_void_p))
        if kind in ("f64", "both"):
            a.frames_f64 = C.cast(ptrs, C.POINTER(C.c_void_p))
        if rng.random() < 0.5:
            a.stds = C.cast(ptrs, C.POINTER(C.c_void_p))
        exp = (C.c_double * nn)(*[rng.choice([1e-3, 1.0, 0.0, -1.0, float("nan"), float("inf"), 1e-310]) for _ in range(nn)])
        if rng.random() < 0.9:
            a.exposures = C.cast(exp, C.POINTER(C.c_double))
        for f in ("icrf", "icrf_diff", "w_lut", "dw_lut", "flat_u8", "flat_f64", "flat_std", "out_val", "out_std", "out_sum_w", "hot_workspace",
                  "frames_workspace"):
            if rng.random() < 0.6:
                setattr(a, f, fake())
        keep = [ptrs, exp]
        if rng.random() < 0.3:
            a.darks_u8 = C.cast(ptrs, C.POINTER(C.c_void_p))
            dm = (C.c_int32 * nn)(*[rng.choice([1, 100, 256, 0, -5, 300]) for _ in range(nn)])
            keep.append(dm)
            if rng.random() < 0.8:
                a.dark_min_dn = C.cast(dm, C.POINTER(C.c_int32))
            a.median_k = rng.choice([3, 5, 7, 1, 2, 4, 9, 0, -3])
        a.hot_workspace_bytes = rng.choice([0, 16, 1 << 20, 0xFFFFFFFF])
        a.frames_workspace_bytes = rng.choice([0, 16, 1 << 30])
        rc = lib.hm_merge(C.byref(a), None)
        assert rc <= 0
        seen[rc] = seen.get(rc, 0) + 1
        buf = C.create_string_buffer(256)
        assert lib.hm_merge_describe(C.byref(a), buf, 256) <= 0
        lib.hm_merge_algorithmic_bytes(C.byref(a))
    assert seen.get(nat.HM_EINVAL, 0) > 1000 and len(seen) >= 3, seen


def test_every_entry_point_survives_random_arguments():
    """8 000 calls of random entry points of the HIP library with random scalars (negative, zero, 2^31 + 7, 2^40, 2^62 element counts; channel
    counts 0..1000; NaN / inf parameters), fake or NULL device pointers and valid host arrays, on a machine WITHOUT a GPU: every int-returning
    call returns <= 0 (HM_OK only where there is nothing to do, HM_ELAUNCH where valid arguments meet no device) and nothing crashes - no
    integer overflow in launch arithmetic (found by this test: hm_axis_statistics_workspace_bytes(12, 2^31 + 7, 2^62) divided by zero),
    no dereference of a device pointer on the host. Host arrays are 4096 entries long and counts that index them stay below that."""
    import random
    from camera_linearity_amd import _native as nat
    lib = nat.hip_lib
    rng = random.Random(3)
    skip = {"hm_version", "hm_strerror", "hm_device_info", "hm_debug_clock_probe", "hm_debug_copy_probe", "hm_debug_stride_probe",
            "hm_gaussian_weight_lut_host", "hm_merge", "hm_merge_algorithmic_bytes", "hm_merge_describe", "hm_tiff_lzw_decode", "hm_tiff_packbits_decode"}

    def fake():
        return 0x7f0000000000 + rng.randrange(0, 1 << 20) * 16 + rng.choice([0, 0, 0, 1, 4, 8])
    keep = []

    def gen(t):
        if t is C.c_void_p:
            return fake() if rng.random() < 0.85 else None
        if t is C.c_int64:
            return rng.choice([-1, 0, 1, 2, 3, 5, 12, 64, 1000, 4096, 1 << 20, (1 << 31) + 7, 1 << 40, 1 << 62])
        if t is C.c_int:
            return rng.choice([-1, 0, 1, 2, 3, 4, 5, 7, 8, 9, 16, 32, 33, 255, 256, 1000])
        if t is C.c_double:
            return rng.choice([0.0, 1.0, -1.0, 0.5, 2.2, 1e-310, float("nan"), float("inf")])
        if isinstance(t, type) and issubclass(t, C._Pointer):
            if rng.random() < 0.1:
                return None
            et = t._type_
            if et is C.c_void_p:
                arr = (C.c_void_p * 4096)(*[fake() if rng.random() < 0.9 else None for _ in range(4096)])
            elif et is C.c_double:
                arr = (C.c_double * 4096)(*[rng.choice([1e-3, 1.0, 0.0, -1.0, 2.0]) for _ in range(4096)])
            elif et is C.c_int32:
                arr = (C.c_int32 * 4096)(*[rng.choice([0, 1, 2, 5, -1, 40]) for _ in range(4096)])
            elif et is C.c_int64:
                arr = (C.c_int64 * 4096)(*[rng.choice([0, 1, 2, 3, 7, -1, 1 << 33]) for _ in range(4096)])
            else:
                return None
            keep.append(arr)
            return C.cast(arr, t)
        raise TypeError(t)
    names = sorted(n for n in nat._SIGNATURES if n not in skip)
    codes = set()
    for _ in range(8000):
        name = rng.choice(names)
        res, argt = nat._SIGNATURES[name]
        a = [gen(t) for t in argt]
        if name == "hm_pairs_statistics":                  # lower / upper are HOST arrays of C doubles (typed void* in the table)
            a[9], a[10] = gen(C.POINTER(C.c_double)), gen(C.POINTER(C.c_double))
        del keep[:-16]
        rc = getattr(lib, name)(*a)
        if res is C.c_int:
            assert rc <= 0, (name, rc)
            codes.add(rc)
    assert nat.HM_EINVAL in codes and nat.HM_ELAUNCH in codes
