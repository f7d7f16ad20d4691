"""The reference's own unit / integration test cases, restated against BOTH backends of this package.

Same classes and test names as the reference's suite:
  tests/unit/test_measurand.py            :120-522   TestMeasurandInitialization ... TestMeasurandApplyThreshold
  tests/unit/test_image_set.py            :107-345   TestImageSetInitialization, TestImageSetMockMeasurand, TestImageSetIO, TestSupportFunctions
  tests/unit/test_exposure_series.py      :26-110    TestExposureSeriesInitialization, TestExposureSeriesFromImageSet
  tests/unit/test_general_functions.py    :10-37     test_is_broadcastable
  tests/integration/test_integration_image_set.py :34-83   init with val/std, 8-bit and 64-bit save / load round trips
so a maintainer can diff behaviour class by class. The reference mocks cv2 and its Measurand for the ImageSet tests;
here the real Measurand classes and the real TIFF codec run instead (nothing to mock: there is no cv2). Tolerances are the
reference's (atol 1e-8 for the algebra). EVERY test runs twice (module-scoped `backend` fixture): on the HIP backend
(`use_cupy=True`, the reference's CuPy slot; gpu marker, runs on the MI355X) and on the host backend (`use_cupy=False`, the
reference's NumpyMeasurand slot backed by libhdrmerge_host.so; runs in the CPU suite).
"""
from copy import deepcopy
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

torch = pytest.importorskip("torch")
ATOL = 1.e-8
BACKEND = {"name": "numpy"}


@pytest.fixture(scope="module", autouse=True, params=[pytest.param("hip", marks=pytest.mark.gpu), "numpy"])
def backend(request):
    if request.param == "hip" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    BACKEND["name"] = request.param
    yield request.param
    BACKEND["name"] = "numpy"


def use_cupy():
    return BACKEND["name"] == "hip"


def M(val=None, std=None):
    from camera_linearity_amd.measurand_factory import Measurand
    return Measurand(val, std, use_cupy=use_cupy())


def IS(**kw):
    from camera_linearity_amd.image_set import ImageSet
    return ImageSet(use_cupy=use_cupy(), **kw)


def ARR():
    return torch.Tensor if use_cupy() else np.ndarray


def host(t):
    if t is None or isinstance(t, np.ndarray):
        return t
    return t.cpu().numpy()


# ---- strategies: two broadcast-compatible float64 arrays in (0, 1], optional std = 0.1 * value (test_measurand.py:26-78)
@st.composite
def broadcastable_measurands(draw, max_dims=5, max_side=10):
    n1, n2 = draw(st.integers(1, max_dims)), draw(st.integers(1, max_dims))
    s1 = [draw(st.integers(1, max_side)) for _ in range(n1)]
    s2 = [draw(st.integers(1, max_side)) for _ in range(n2)]
    top = max(n1, n2)
    p1, p2 = [1] * (top - n1) + s1, [1] * (top - n2) + s2
    for i in range(top):                                    # make every axis pair equal or 1
        if p1[i] != p2[i] and p1[i] != 1 and p2[i] != 1:
            if p1[i] > p2[i]:
                p2[i] = 1
            else:
                p1[i] = 1
    rng = np.random.default_rng(draw(st.integers(0, 2 ** 32 - 1)))
    a, b = rng.random(p1), rng.random(p2)
    a, b = a / a.max(), b / b.max()
    sa = 0.1 * a if draw(st.booleans()) else None
    sb = 0.1 * b if draw(st.booleans()) else None
    return M(a, sa), M(b, sb)


def std_rule(r1, r2, m1, m2, compare=True):
    """Results carry a std iff either operand does (measurand.py:281-302)."""
    if m1.std is not None or m2.std is not None:
        assert r1.std is not None and r2.std is not None
        if compare:
            assert np.allclose(host(r1.std), host(r2.std), atol=ATOL)
    else:
        assert r1.std is None and r2.std is None


def same_as(result, m):
    assert np.allclose(host(result.val), np.broadcast_to(host(m.val), result.val.shape), atol=ATOL)
    if m.std is not None:
        assert np.allclose(host(result.std), host(m.std), atol=ATOL)
    else:
        assert result.std is None


prop = settings(deadline=None, max_examples=20)


# ================================================================ tests/unit/test_measurand.py
class TestMeasurandInitialization:                                   # :120-167 (CPU: no arithmetic)
    def test_initialize_with_float_value(self):
        m = M(10.0)
        assert isinstance(m.val, ARR()) and str(m.val.dtype).endswith("float64")
        assert float(m.val) == 10.0 and m.std is None

    def test_initialize_with_float_value_and_std(self):
        m = M(10.0, 1.0)
        assert float(m.val) == 10.0 and float(m.std) == 1.0 and str(m.std.dtype).endswith("float64")

    def test_initialize_with_array_value(self):
        v = np.array([10.0, 20.0])
        m = M(v)
        assert np.array_equal(host(m.val), v) and m.std is None

    def test_initialize_with_array_value_and_std(self):
        v, s = np.array([10.0, 20.0]), np.array([1.0, 2.0])
        m = M(v, s)
        assert isinstance(m.val, ARR()) and np.array_equal(host(m.val), v)
        assert isinstance(m.std, ARR()) and np.array_equal(host(m.std), s)

    def test_initialize_with_invalid_value_type(self):
        with pytest.raises(TypeError, match="Invalid value type"):
            M("invalid_val", 1.0)

    def test_initialize_with_invalid_std_type(self):
        with pytest.raises(TypeError, match="Invalid std type"):
            M(10.0, "invalid_std")

    def test_initialize_with_none_std(self):
        m = M(10.0, None)
        assert float(m.val) == 10.0 and m.std is None


class TestMeasurandAddition:                                         # :170-208
    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_commutativity(self, ms):
        a, b = ms
        r1, r2 = a + b, b + a
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_associativity(self, ms):
        a, b = ms
        c = deepcopy(a)
        r1, r2 = (a + b) + c, (b + c) + a
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_identity(self, ms):
        same_as(ms[0] + 0, ms[0])


class TestMeasurandSubtraction:                                      # :211-245
    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_commutativity(self, ms):
        a, b = ms
        r1, r2 = a - b, b - a
        assert np.allclose(host(r1.val), -1 * host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_self_subtraction(self, ms):
        a, _ = ms
        r = a - a
        assert np.allclose(host(r.val), 0.0, atol=ATOL)
        if a.std is not None:
            assert bool((r.std >= a.std).all())
        else:
            assert r.std is None

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_identity(self, ms):
        same_as(ms[0] - 0, ms[0])


class TestMeasurandDivision:                                         # :248-310
    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_inversion(self, ms):
        a, b = ms
        r1, r2 = a / b, b / a
        assert np.allclose(host(r1.val), 1 / host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b, compare=False)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_distributivity(self, ms):
        a, b = ms
        c = deepcopy(a)
        r1, r2 = (a + b) / c, a / c + b / c
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b, compare=False)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_identity(self, ms):
        same_as(ms[0] / 1, ms[0])

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_unity_property(self, ms):
        a, _ = ms
        r = a / a
        assert np.allclose(host(r.val), 1.0, atol=ATOL)
        assert (r.std is not None) == (a.std is not None)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_division_by_zero(self, ms):
        a, _ = ms
        r = a / 0
        assert bool(np.isinf(host(r.val)).all()) and bool((host(r.val) > 0).all())
        assert (r.std is not None) == (a.std is not None)


class TestMeasurandMultiplication:                                   # :313-378
    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_commutativity(self, ms):
        a, b = ms
        r1, r2 = a * b, b * a
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_associativity(self, ms):
        a, b = ms
        c = deepcopy(a)
        r1, r2 = (a * b) * c, (b * c) * a
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_distributivity(self, ms):
        a, b = ms
        c = deepcopy(a)
        r1, r2 = a * (b + c), (a * b) + (a * c)
        assert np.allclose(host(r1.val), host(r2.val), atol=ATOL)
        std_rule(r1, r2, a, b, compare=False)

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_identity(self, ms):
        same_as(ms[0] * 1, ms[0])

    @prop
    @given(broadcastable_measurands())
    def test_broadcastable_measurand_zero_property(self, ms):
        a, _ = ms
        r = a * 0
        assert np.allclose(host(r.val), 0.0, atol=ATOL)
        if a.std is not None:
            assert np.allclose(host(r.std), 0.0, atol=ATOL)
        else:
            assert r.std is None


class TestNormalizeInput:                                            # :381-444 (CPU)
    def test_other_is_measurand(self):
        m1, m2 = M(10.0), M(20.0)
        other, use_std = m1._normalize_input(m2)
        assert other is m2 and use_std is False

    def test_other_is_measurand_with_std(self):
        m1, m2 = M(10.0, 1.0), M(20.0, 2.0)
        other, use_std = m1._normalize_input(m2)
        assert other is m2 and use_std is True

    def test_other_is_float(self):
        other, use_std = M(10.0, 1.0)._normalize_input(20.0)
        assert type(other) is type(M(1.0)) and float(other.val) == 20.0 and other.std is None and use_std is True

    def test_other_is_cnp_array(self):
        other, use_std = M(10.0, 1.0)._normalize_input(np.array([1, 2, 3]))
        assert type(other) is type(M(1.0)) and np.array_equal(host(other.val), [1, 2, 3])
        assert other.std is None and use_std is True

    def test_invalid_other_type_raises_type_error(self):
        with pytest.raises(TypeError, match="Invalid other type."):
            M(10.0)._normalize_input("invalid_string")

    def test_other_is_measurand_with_std_and_use_std(self):
        m1, m2 = M(10.0, 1.0), M(20.0, 2.0)
        other, use_std = m1._normalize_input(m2)
        assert other is m2 and use_std is True

    def test_other_is_xp_array_and_use_std(self):
        other, use_std = M(10.0, 1.0)._normalize_input(torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64))
        assert np.array_equal(host(other.val), [1, 2, 3]) and other.std is None and use_std is True


class TestMeasurandLinearize:                                        # :447-467 (fails at reference HEAD; holds here)
    @prop
    @given(broadcastable_measurands())
    def test_linearize(self, ms):
        m, _ = ms
        channels = m.val.shape[-1]
        if channels > 4 and use_cupy():                     # HM_MAX_CHANNELS of the device kernels (the host build takes any number)
            with pytest.raises((ValueError, NotImplementedError)):
                m.linearize(np.zeros((256, channels)))
            return
        icrf = np.stack([np.linspace(0, 1, 256) ** (c + 1) for c in range(channels)], axis=1)
        diff = np.stack([np.gradient(icrf[:, c], 2 / 255) for c in range(channels)], axis=1)
        lin = m.linearize(icrf[:, 0], diff[:, 0]) if channels == 1 else m.linearize(icrf, diff)
        assert tuple(lin.val.shape) == tuple(m.val.shape)
        for c in range(channels):
            assert np.isin(host(lin.val[..., c]), icrf[:, c]).all()
        assert (lin.std is not None) == (m.std is not None)


class TestMeasurandApplyThreshold:                                   # :470-522
    @prop
    @given(broadcastable_measurands(), st.floats(0.25, 0.75), st.integers(0, 2 ** 31))
    def test_apply_thresholds_regression(self, ms, threshold, seed):
        m, _ = ms
        n = m.val.shape[-1]
        rng = np.random.default_rng(seed)
        lower = [None if rng.random() < 0.3 else float(rng.uniform(0.0, threshold)) for _ in range(n)]
        upper = [None if rng.random() < 0.3 else float(rng.uniform(threshold, 1.0)) for _ in range(n)]
        val0, std0 = host(m.val).copy(), None if m.std is None else host(m.std).copy()
        t = deepcopy(m)
        t.apply_thresholds(lower, upper)
        for c in range(n):                                  # the simple per-channel masking the reference regresses against
            lo = -np.inf if lower[c] is None else lower[c]
            hi = np.inf if upper[c] is None else upper[c]
            mask = (val0[..., c] < lo) | (val0[..., c] > hi)
            val0[..., c][mask] = np.nan
            if std0 is not None:
                std0[..., c][mask] = np.nan
        assert np.allclose(host(t.val), val0, equal_nan=True)
        if std0 is not None:
            assert np.allclose(host(t.std), std0, equal_nan=True)


# ================================================================ tests/unit/test_general_functions.py
@given(st.lists(st.integers(min_value=1), min_size=0, max_size=10), st.lists(st.integers(min_value=1), min_size=0, max_size=10))
def test_is_broadcastable(shape1, shape2):                           # :10-37
    from camera_linearity_amd.measurand import is_broadcastable
    if not shape1 or not shape2:
        with pytest.raises(ValueError, match="Shapes cannot be empty"):
            is_broadcastable(shape1, shape2)
        return
    top = max(len(shape1), len(shape2))
    a, b = [1] * (top - len(shape1)) + shape1, [1] * (top - len(shape2)) + shape2
    expected = all(x == 1 or y == 1 or x == y for x, y in zip(a, b))
    assert is_broadcastable(shape1, shape2) == expected


# ================================================================ tests/unit/test_image_set.py
class TestImageSetInitialization:                                    # :107-147 (CPU)
    def test_imageset_init_with_no_args(self):
        from camera_linearity_amd.image_set import ImageSet
        s = IS()
        assert s.measurand is not None and s.measurand.val is None and s.measurand.std is None
        assert s.path is None and s.features is None

    def test_imageset_init_with_mock_measurand(self):
        from camera_linearity_amd.image_set import ImageSet
        m = M()
        assert IS(measurand=m).measurand is m

    def test_multiple_from_path_mock_glob(self):
        from camera_linearity_amd.image_set import ImageSet
        mock_path = MagicMock(spec=Path)
        mock_path.glob.side_effect = lambda pat: [Path("image1.tif"), Path("image2_STD.tif"), Path("image3.tif")] if pat == "*.tif" else []
        sets = ImageSet.multiple_from_path(mock_path)
        assert [s.path for s in sets] == [Path("image1.tif"), Path("image3.tif")]
        assert all(isinstance(s, ImageSet) for s in sets)


class TestImageSetMockMeasurand:                                     # :150-216, with real measurands
    def test_imageset_linearize(self):
        from camera_linearity_amd.image_set import ImageSet
        icrf = np.stack([np.linspace(0, 1, 256) ** 2.0] * 3, axis=1)
        v = np.random.default_rng(0).integers(0, 256, (5, 6, 3), dtype=np.uint8)
        out = IS(value=v).linearize(icrf)
        assert isinstance(out, ImageSet) and np.array_equal(host(out.measurand.val), icrf[v, np.arange(3)])

    def test_imageset_compute_difference(self, tmp_path):
        from camera_linearity_amd.image_set import ImageSet
        rng = np.random.default_rng(1)
        a, b = rng.random((4, 5, 3)) + 0.1, rng.random((4, 5, 3)) + 0.1
        s1 = IS(value=a, file_path=tmp_path / "20ms test_image.tif")
        s2 = IS(value=b, file_path=tmp_path / "50ms test_image.tif")
        ab, rel = ImageSet.compute_difference(s1, s2)                 # ratio 20 / 50 from the file names
        scaled = b * (0.02 / 0.05)
        assert np.allclose(host(ab.measurand.val), a - scaled, atol=ATOL)
        assert np.allclose(host(rel.measurand.val), (a - scaled) / scaled, atol=ATOL)

    def test_imageset_exposure_interpolation(self, tmp_path):
        from camera_linearity_amd.image_set import ImageSet
        rng = np.random.default_rng(2)
        a, b = rng.random((4, 5, 3)), rng.random((4, 5, 3))
        s1 = IS(value=a, file_path=tmp_path / "20ms test_image.tif")
        s2 = IS(value=b, file_path=tmp_path / "50ms test_image.tif")
        r = ImageSet.exposure_interpolation(s1, s2, 0.04)
        assert np.allclose(host(r.measurand.val), a + (b - a) * (0.04 - 0.02) / (0.05 - 0.02), atol=ATOL)
        with pytest.raises(ValueError):
            ImageSet.exposure_interpolation(s1, s2, 0.06)
        with pytest.raises(TypeError):
            ImageSet.exposure_interpolation(s1, s2, 1)

    def test_imageset_extract(self):
        from camera_linearity_amd.image_set import ImageSet
        v = np.random.default_rng(3).random((4, 5, 3))
        r = IS(value=v, std=0.1 * v).extract([0, 2])
        assert np.array_equal(host(r.measurand.val), v[..., [0, 2]]) and np.array_equal(host(r.measurand.std), 0.1 * v[..., [0, 2]])


class TestImageSetIO:                                                # :219-312, against real files instead of a patched cv2.imread
    def test_load_value_image_8bit(self, tmp_path):
        from camera_linearity_amd import tiff_io
        from camera_linearity_amd.image_set import ImageSet
        img = np.full((3, 3, 3), 128, dtype=np.uint8)
        tiff_io.imwrite(tmp_path / "image.tif", img)
        s = IS(file_path=tmp_path / "image.tif")
        s.load_value_image(bit64=False)
        np.testing.assert_array_equal(host(s.measurand.val), img.astype(np.float64) / 255)

    def test_load_value_image_64bit(self, tmp_path):
        from camera_linearity_amd import tiff_io
        from camera_linearity_amd.image_set import ImageSet
        img = np.full((3, 3, 3), 128, dtype=np.uint8)
        tiff_io.imwrite(tmp_path / "image.tif", img)
        s = IS(file_path=tmp_path / "image.tif")
        s.load_value_image(bit64=True)                                # raw values, no / MAX_DN (image_set.py:225)
        np.testing.assert_allclose(host(s.measurand.val), img.astype(np.float64))

    def test_load_std_image_not_found(self, tmp_path):
        from camera_linearity_amd.image_set import ImageSet
        s = IS(file_path=tmp_path / "image.tif", value=np.zeros((3, 3, 3), np.uint8))
        calls = []
        s.calculate_numerical_STD = lambda data=None: calls.append(data)     # returns None -> std stays unset
        s.load_std_image(bit64=True)
        assert len(calls) == 1 and s.measurand.std is None

    def test_load_std_image_found(self, tmp_path):
        from camera_linearity_amd import tiff_io
        from camera_linearity_amd.image_set import ImageSet
        std = np.ones((3, 3, 3), dtype=np.float64)
        tiff_io.imwrite(tmp_path / "image STD.tif", std)
        s = IS(file_path=tmp_path / "image.tif", value=np.zeros((3, 3, 3), np.uint8))
        s.load_std_image(bit64=True)
        np.testing.assert_allclose(host(s.measurand.std), std)

    def test_calculate_numerical_STD(self):
        from camera_linearity_amd.image_set import ImageSet
        table = np.stack([np.linspace(0, 1, 256) ** 2.0] * 3, axis=1)      # same format as an ICRF (test_image_set.py:295)
        s = IS(file_path=Path("dummy/path/image.tif"), value=np.ones((3, 3, 3)) / 100)
        result = host(s.calculate_numerical_STD(table))
        for c in range(3):
            assert np.isin(result[..., c], table[..., c]).all()

    def test_calculate_numerical_STD_not_found(self):
        from camera_linearity_amd.image_set import ImageSet
        assert IS(file_path=Path("dummy/path/image.tif")).calculate_numerical_STD() is None


class TestSupportFunctions:                                          # :315-345 (CPU)
    @pytest.mark.parametrize("file_path, expected_features", [
        (Path("bf 40x 100ms sample.tif"), {"illumination": "bf", "magnification": "40x", "exposure": 0.1, "subject": "sample"}),
        (Path("df 20x 200ms test_subject.tif"), {"illumination": "df", "magnification": "20x", "exposure": 0.2, "subject": "test_subject"}),
        (Path("40x 500ms subject1.tif"), {"illumination": "", "magnification": "40x", "exposure": 0.5, "subject": "subject1"}),
        (Path("bf 1000ms.tif"), {"illumination": "bf", "magnification": "", "exposure": 1.0, "subject": ""}),
        (Path("sample 10x.tif"), {"illumination": "", "magnification": "10x", "exposure": 0, "subject": "sample"}),
    ])
    def test_features_from_file_name(self, file_path, expected_features):
        from camera_linearity_amd.image_set import _features_from_file_name
        assert _features_from_file_name(file_path) == expected_features

    def test_is_exposure_match(self):
        from camera_linearity_amd.image_set import ImageSet
        f = {"illumination": "bf", "magnification": "40x", "subject": "sample"}
        assert IS(features=dict(f)).is_exposure_match(IS(features=dict(f))) is True
        assert IS(features=dict(f)).is_exposure_match(IS(features=dict(f, magnification="20x"))) is False
        assert IS(features=dict(f, exposure=0.1)).is_exposure_match(IS(features=dict(f, exposure=0.2))) is True


# ================================================================ tests/unit/test_exposure_series.py
def _mock_sets():
    from camera_linearity_amd.image_set import ImageSet
    out = []
    for e in (100, 200, 50):
        s = MagicMock(spec=ImageSet)
        s.features = {"exposure": e}
        s.use_cupy = use_cupy()
        out.append(s)
    return out


class TestExposureSeriesInitialization:                              # :26-89 (CPU)
    def test_initialization_with_all_args(self, tmp_path):
        from camera_linearity_amd.exposure_series import ExposureSeries
        s1, s2, _ = _mock_sets()
        d = tmp_path / "mock" / "directory"
        es = ExposureSeries(merged_image_set=s1, directory_path=d, input_image_sets=[s1, s2])
        assert es.merged_image_set is s1 and es.directory_path == d and es.input_image_sets == [s1, s2] and es.exposure_pairs is None

    def test_initialization_with_only_merged_image_set(self):
        from camera_linearity_amd.exposure_series import ExposureSeries
        s1, _, _ = _mock_sets()
        es = ExposureSeries(merged_image_set=s1)
        assert es.merged_image_set is s1 and es.input_image_sets == [] and es.directory_path is None and es.exposure_pairs is None

    def test_initialization_with_directory_path(self, tmp_path):
        from camera_linearity_amd.exposure_series import ExposureSeries
        f = tmp_path / "mock" / "directory" / "some_file.tif"
        assert ExposureSeries(directory_path=f).directory_path == f.parent

    def test_initialization_with_none(self):
        from camera_linearity_amd.exposure_series import ExposureSeries
        es = ExposureSeries()
        assert es.merged_image_set is None and es.input_image_sets == [] and es.directory_path is None and es.exposure_pairs is None

    def test_initialization_with_directory_path_as_directory(self):
        from camera_linearity_amd.exposure_series import ExposureSeries
        d = Path("/mock/directory")
        assert ExposureSeries(directory_path=d).directory_path == d


class TestExposureSeriesFromImageSet:                                # :92-110 (CPU)
    def test_from_image_set(self, monkeypatch):
        from camera_linearity_amd.exposure_series import ExposureSeries
        from camera_linearity_amd.image_set import ImageSet
        s1, s2, s3 = _mock_sets()
        monkeypatch.setattr(ImageSet, "multiple_from_path", classmethod(lambda cls, path, use_cupy=False: [s1, s2, s3]))
        ref = MagicMock(spec=ImageSet)
        ref.path = Path("/fake/path/reference_image_set.tif")
        ref.features = {"exposure": 150}
        ref.is_exposure_match.return_value = True
        es = ExposureSeries.from_image_set(ref)
        assert [s.features["exposure"] for s in es.input_image_sets] == [50, 100, 200]      # sorted by exposure (:143)
        assert es.directory_path == ref.path.parent


# ================================================================ tests/integration/test_integration_image_set.py
def _random_array(shape=(100, 100, 3), lo=0.0, hi=1.0, seed=0):
    return np.random.default_rng(seed).random(shape) * (hi - lo) + lo


class TestImageSetInitializationIntegration:                         # :34-43
    def test_imageset_init_with_val_and_std(self):
        from camera_linearity_amd.image_set import ImageSet
        v = _random_array()
        s = IS(value=v, std=v * 0.1)
        assert np.all(host(s.measurand.val) == v) and np.all(host(s.measurand.std) == v * 0.1)


class TestImageSetIOIntegration:                                     # :46-83
    def test_imageset_save_and_load_8bit(self, tmp_path):
        from camera_linearity_amd.image_set import ImageSet
        full_path = tmp_path / "1.0ms test_image BF 5x.tif"
        v = _random_array()
        s = IS(file_path=full_path, value=v, std=v * 0.1)
        s.save_8bit(save_path=full_path)
        other = IS(file_path=full_path)
        other.load_value_image(bit64=False)
        other.load_std_image(bit64=False)
        assert np.allclose(host(s.measurand.val), host(other.measurand.val), atol=0.5 / 255)
        assert np.allclose(host(s.measurand.std), host(other.measurand.std))

    def test_imageset_save_and_load_64bit(self, tmp_path):
        from camera_linearity_amd.image_set import ImageSet
        full_path = tmp_path / "1.0ms test_image BF 5x.tif"
        v = _random_array(seed=1)
        s = IS(file_path=full_path, value=v, std=v * 0.1)
        s.save_64bit(save_path=full_path)
        other = IS(file_path=full_path)
        other.load_value_image(bit64=True)
        other.load_std_image(bit64=True)
        assert np.array_equal(host(s.measurand.val), host(other.measurand.val))
        assert np.array_equal(host(s.measurand.std), host(other.measurand.std))
