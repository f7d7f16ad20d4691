"""camera_linearity_amd.general_functions - the reference's host-side helpers (modules/general_functions.py) - against outputs the
reference itself produced (tests/golden/helpers.npz, written by tests/golden/make_golden.py `helpers`). CPU only."""
import numpy as np
import pytest

gf = pytest.importorskip("camera_linearity_amd.general_functions")


def test_shape_helpers(golden):
    g = golden("helpers")
    for a, la, b, lb, want in zip(g["bc_shapes_a"], g["bc_len_a"], g["bc_shapes_b"], g["bc_len_b"], g["bc_result"]):
        assert gf.is_broadcastable(tuple(int(x) for x in a[:la]), tuple(int(x) for x in b[:lb])) == bool(want)
    with pytest.raises(ValueError):
        gf.is_broadcastable((), (1,))
    img = g["ces_in"]
    assert np.array_equal(gf.choose_evenly_spaced_points(img, 5), g["ces_5"])
    assert np.array_equal(gf.choose_evenly_spaced_points(img, 4, 7), g["ces_4_7"])
    got = [gf.predict_output_shape((23, 31), 5), gf.predict_output_shape((23, 31), 4, 7), gf.predict_output_shape((1, 1), 9)]
    assert np.array_equal(np.array(got), g["pos"])
    assert gf.predict_output_shape((23, 31), 5) == gf.choose_evenly_spaced_points(img, 5).shape[:2]


def test_weighted_statistics(golden):
    g = golden("helpers")
    np.testing.assert_allclose(np.array(gf.weighted_avg_and_std(g["was_v"], g["was_w"])), g["was"], rtol=1e-14)
    np.testing.assert_allclose(np.array(gf.weighted_avg_and_std(g["was_v"], None)), g["was_none"], rtol=1e-14)
    for axis, key in ((0, "na_axis0"), ((0, 1), "na_axis01"), (2, "na_axis2")):
        got = gf.nanaverage(g["na_v"], g["na_w"], axis)
        assert np.array_equal(np.isnan(got), np.isnan(g[key])), key
        np.testing.assert_allclose(got, g[key], rtol=1e-14, equal_nan=True)
    assert np.isnan(g["na_axis0"]).any()                       # the line without a valid weight is in the fixture
    np.testing.assert_allclose(gf.weighted_percentile(g["wp_v"]), g["wp_default"], rtol=1e-14)
    np.testing.assert_allclose(gf.weighted_percentile(g["wp_v"], np.array([5.0, 50.0, 95.0]), g["wp_w"]), g["wp_weighted"], rtol=1e-14)


def test_map_linearity_limits_and_icrf_file(golden, tmp_path):
    from camera_linearity_amd import settings
    g = golden("helpers")
    icrf = g["mll_icrf"]
    settings.configure(LOWER_LIN_LIM=5, UPPER_LIN_LIM=250)
    for args, key in (((None, None, icrf), "mll_none_icrf"), ((10, 20, icrf), "mll_10_20_icrf"), ((None, None, None), "mll_none_none"),
                      ((7, 3, None), "mll_7_3_none")):
        lo, hi = gf.map_linearity_limits(*args)
        assert np.array_equal(np.stack([lo, hi]), g[key]), key
    p = tmp_path / "ICRF_calibrated.txt"
    np.savetxt(p, icrf)
    a, d = gf.read_ICRF_file(p)
    np.testing.assert_allclose(a, icrf, rtol=1e-15)
    np.testing.assert_allclose(d[:, 1], np.gradient(a[:, 1], 2 / 255), rtol=1e-15)       # dx = 2 / (BITS - 1), general_functions.py:270
    assert gf.read_ICRF_file(p, return_derivative=False)[1] is None
    np.testing.assert_allclose(gf.read_txt_to_array("ICRF_calibrated.txt", str(tmp_path)), icrf, rtol=1e-15)
    with pytest.raises(ImportError, match="frames_from_capture"):
        next(gf.video_frame_generator(tmp_path / "missing.avi"))


def test_global_settings_class_is_a_view_of_settings():
    from camera_linearity_amd import settings
    from camera_linearity_amd.global_settings import GlobalSettings as gs
    assert gs.BITS == 256 and gs.MAX_DN == 255 and gs.NUM_OF_CHS == 3 and gs.CH_STR[2] == "Red"
    old = settings.DARK_THRESHOLD
    try:
        settings.configure(DARK_THRESHOLD=0.123)
        assert gs.DARK_THRESHOLD == 0.123                   # read at access time
        gs.DARK_THRESHOLD = 0.2                             # assignment configures
        assert settings.DARK_THRESHOLD == 0.2
        with pytest.raises(NotImplementedError):
            gs.BIT_DEPTH = 12                               # the kernels are built for 8-bit DNs
    finally:
        settings.configure(DARK_THRESHOLD=old)
    with pytest.raises(AttributeError, match="PCA_FILES"):
        gs.PCA_FILES
    assert "LOWER_LIN_LIM" in dir(gs)
