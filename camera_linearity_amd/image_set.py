"""ImageSet - mirror of modules/image_set.py:25-568 on the HIP backend.

One image = a path, the features parsed from its file name, and one HipMeasurand. Method names,
arguments and return values follow the reference; the arithmetic they trigger runs in the HIP
kernels behind HipMeasurand. File naming grammar (image_set.py:1-9, 542-568): descriptors separated
by spaces - '<exposure>ms', 'bf'/'df', '<magnification>x', subject; ' STD' marks an uncertainty image.

IO: the reference reads/writes TIFF through OpenCV. Here `load_value_image` /
`load_std_image` / `save_64bit` / `save_8bit` go through `tiff_io` (cv.imread / cv.imwrite conventions
without cv2: BGR arrays, RGB files, 8-bit and float64 samples) and also accept `.npy` arrays; images can
always be supplied in memory (`value=`, `std=`, `measurand=`). 8-bit images are kept as uint8 DNs in HBM.

Deviations: G - bad_pixel_filter / flat_field_correction return the new ImageSet as
the reference does, and the merge loop uses the result; I - scale_to_exposure scales by
target/original exposure and leaves the source's features untouched; J - works on in-memory images.
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import settings as gs
from . import tiff_io
from .measurand import HipMeasurand, HostMeasurand
from .measurand_factory import Measurand, measurand_to_cupy, measurand_to_numpy


def _read_image(path: Path, unchanged: bool = False) -> Optional[np.ndarray]:
    path = Path(path)
    if path.suffix == ".npy":
        return np.load(path, allow_pickle=False) if path.exists() else None
    return tiff_io.imread(path, tiff_io.IMREAD_UNCHANGED if unchanged else tiff_io.IMREAD_COLOR)     # cv.imread's contract


def _std_path(path: Path) -> Path:
    s = str(path)
    suffix = Path(s).suffix
    return Path(s.removesuffix(suffix) + " STD" + suffix)     # image_set.py:235


class ImageSet(object):

    def __init__(self, file_path=None, value=None, std=None, features: Optional[Dict] = None,
                 measurand: Optional[HipMeasurand] = None, use_cupy: Optional[bool] = False):
        """image_set.py:27-45, defaults included: `use_cupy=False` is the host backend (the reference's NumPy slot), `use_cupy=True`
        the device backend (HIP in the reference's CuPy slot); a given measurand decides for itself."""
        self.path = Path(file_path) if isinstance(file_path, str) else file_path
        if measurand is not None:                       # image_set.py:37-42
            if getattr(measurand, "backend", None) not in ("hip", "numpy"):
                raise ValueError(f"Expected a Measurand of this package, got {type(measurand)} instead.")
            self._measurand = measurand
            self._use_cupy = measurand.backend != "numpy"
        else:
            self._use_cupy = bool(use_cupy)
            if isinstance(value, np.ndarray) and value.dtype == np.uint8 or \
                    isinstance(value, torch.Tensor) and value.dtype == torch.uint8:
                self._measurand = self._measurand_class().from_dn(value, std)
            else:
                self._measurand = Measurand(value, std, self._use_cupy)
        if features is not None:
            self.features = features
        elif file_path is not None:
            self.features = _features_from_file_name(self.path)
        else:
            self.features = None
        self.is_HDR = False

    # ---- backend bookkeeping (image_set.py:55-100)
    @property
    def measurand(self):
        return self._measurand

    @measurand.setter
    def measurand(self, new_measurand):
        expected_backend = "hip" if self._use_cupy else "numpy"          # image_set.py:59-77 ("cupy" there)
        if getattr(new_measurand, "backend", None) != expected_backend:
            raise ValueError(f"Expected type {expected_backend}, got {type(new_measurand)} instead.")
        self._measurand = new_measurand

    def _measurand_class(self):
        return HipMeasurand if self._use_cupy else HostMeasurand

    @property
    def use_cupy(self):
        return self._use_cupy

    @use_cupy.setter
    def use_cupy(self, new_value):
        raise AttributeError("use_cupy is a read-only attribute, managing the state of the used array backend.")

    def to_cupy(self):
        """image_set.py:95-100: convert this ImageSet to the device backend (HIP in the CuPy slot)."""
        self._measurand = measurand_to_cupy(self.measurand)
        self._use_cupy = True

    def show_image(self):
        """image_set.py:423-433 opens an OpenCV window; no GUI in this package - use host_arrays() with the viewer of your choice."""
        raise NotImplementedError("show_image needs an OpenCV GUI; use ImageSet.host_arrays() and display the array yourself")

    def to_numpy(self):
        """image_set.py:88-93: convert this ImageSet to the host (NumPy) backend, in place."""
        self._measurand = measurand_to_numpy(self.measurand)
        self._use_cupy = False

    def host_arrays(self):
        """(val, std) of the managed image as NumPy arrays on the host, whatever the backend (a copy from the device on the HIP backend)."""
        return self.measurand.to_numpy()

    # ---- pass-throughs
    def linearize(self, ICRF, ICRF_diff=None):
        """image_set.py:102-115."""
        return ImageSet(file_path=self.path, features=self.features, measurand=self.measurand.linearize(ICRF, ICRF_diff))

    def get_file_path_without_exposure(self):
        if self.path is not None and self.features is not None:
            return self.path.parent.joinpath(
                f"{self.features['subject']} {self.features['illumination']} {self.features['magnification']}.tif")
        return None

    def is_exposure_match(self, other: "ImageSet"):
        """image_set.py:123-144."""
        if self.features is None or other.features is None:
            return False
        for key in self.features.keys():
            if key == "exposure":
                continue
            if self.features[key] != other.features[key]:
                return False
        return True

    def get_flat_field(self, list_of_flat_fields: Optional[List["ImageSet"]] = None):
        """image_set.py:146-155. `None` globs settings.DEFAULT_FLAT_PATH when that is configured (else: no flat field)."""
        if list_of_flat_fields is None and gs.DEFAULT_FLAT_PATH is not None:
            list_of_flat_fields = ImageSet.multiple_from_path(Path(gs.DEFAULT_FLAT_PATH))
        if list_of_flat_fields is None or self.features is None:
            return None
        for flat_set in list_of_flat_fields:
            if self.features["illumination"] == flat_set.features["illumination"] and \
                    self.features["magnification"] == flat_set.features["magnification"]:
                return flat_set
        return None

    def select_dark_field(self, list_of_dark_fields: Optional[List["ImageSet"]], dark_threshold: Optional[float] = None):
        """The selection rule of get_dark_field (image_set.py:171-198) without touching pixels:
        returns (dark ImageSet, scale) or (None, 0.0). scale = target/dark exposure (deviation I)."""
        if list_of_dark_fields is None and gs.DEFAULT_DARK_PATH is not None:      # image_set.py:169-170
            list_of_dark_fields = ImageSet.multiple_from_path(Path(gs.DEFAULT_DARK_PATH))
        if not list_of_dark_fields:
            return None, 0.0
        thr = gs.DARK_THRESHOLD if dark_threshold is None else dark_threshold
        target = self.features["exposure"]
        if not (target >= thr):
            return None, 0.0
        lesser = greater = False
        greater_index = 0
        for i, dark in enumerate(list_of_dark_fields):
            de = dark.features["exposure"]
            if de < target:
                lesser = True
            if de > target:
                greater = True
                greater_index = i
            if target == de:
                return dark, 1.0
            if lesser and greater:
                g = list_of_dark_fields[greater_index]
                return g, target / g.features["exposure"]
        return None, 0.0

    def get_dark_field(self, list_of_dark_fields: Optional[List["ImageSet"]] = None):
        """image_set.py:157-198: the exact-exposure dark, or the next longer one scaled to this exposure."""
        dark, scale = self.select_dark_field(list_of_dark_fields)
        if dark is None:
            return None
        if dark.measurand.shape is None and dark.path is not None:
            dark.load_value_image()
        return dark if scale == 1.0 else dark.scale_to_exposure(self.features["exposure"])

    def extract(self, channels=None):
        return ImageSet(file_path=self.path, features=self.features, measurand=self.measurand.extract(dims=channels, axis=-1))

    def load_value_image(self, bit64: Optional[bool] = False):
        """image_set.py:214-226. 8-bit images stay uint8 DNs on the device; `.measurand.val` is DN/255."""
        img = _read_image(self.path, unchanged=bool(bit64))
        if img is None:
            raise FileNotFoundError(str(self.path))
        std = self.measurand._std
        cls = self._measurand_class()
        if img.dtype == np.uint8 and not bit64:
            self._measurand = cls.from_dn(img, std)
        else:
            self._measurand = cls(img.astype(np.float64), std)

    def load_std_image(self, STD_data=None, bit64: Optional[bool] = False):
        """image_set.py:228-243: '<name> STD.tif' as float64, else the per-DN table fallback."""
        std_array = None
        if self.path is not None:
            std_array = _read_image(_std_path(self.path), unchanged=True)
        if std_array is None:
            std_array = self.calculate_numerical_STD(STD_data)
        if std_array is None:
            return
        self.measurand.std = std_array

    def calculate_numerical_STD(self, STD_data=None):
        """image_set.py:365-385: map DN -> std through a (BITS, C) or (BITS,) table with the linearize kernel."""
        if STD_data is None:
            print("Could not load STD data for numerical estimation.")
            return None
        return self.measurand.linearize(ICRF=STD_data).val

    def scale_to_exposure(self, target_exp: float):
        """image_set.py:245-262 with deviation I."""
        new_features = dict(self.features)
        exposure = self.features["exposure"]
        new_features["exposure"] = target_exp
        new_measurand = (target_exp / exposure) * self.measurand
        return ImageSet(file_path=self.path, features=new_features, measurand=new_measurand)

    def bad_pixel_filter(self, darkSet: "ImageSet", threshold_value: Optional[float] = None):
        """image_set.py:387-400."""
        thr = gs.DARK_THRESHOLD if threshold_value is None else threshold_value
        new_measurand = self.measurand.filter_larger_than_by_map(darkSet.measurand, thr)
        return ImageSet(file_path=self.path, features=self.features, measurand=new_measurand)

    def flat_field_correction(self, flatSet: "ImageSet"):
        """image_set.py:402-421."""
        if flatSet.measurand.shape is None:
            flatSet.load_value_image()
        if flatSet.measurand._std is None and self.measurand._std is not None:
            flatSet.load_std_image()
        new_measurand = self.measurand.normalize_by_map(flatSet.measurand)
        return ImageSet(file_path=self.path, features=self.features, measurand=new_measurand)

    def save_64bit(self, save_path: Optional[Path] = None, is_HDR: Optional[bool] = False,
                   separate_channels: Optional[bool] = False):
        """image_set.py:264-318: float64 TIFFs '<name>[ HDR].tif' and '<name>[ HDR] STD.tif' (or one file per channel,
        '<name>[ HDR] <channel name>.tif')."""
        file_path = self.path.parent.joinpath("64bit", self.path.name) if save_path is None else Path(save_path)
        file_path.parent.mkdir(parents=True, exist_ok=True)
        base = str(file_path).removesuffix(".tif")
        acq_suffix, std_suffix = (" HDR.tif", " HDR STD.tif") if is_HDR else (".tif", " STD.tif")
        val, std = self.host_arrays()
        if not separate_channels:
            tiff_io.imwrite(base + acq_suffix, val.astype(np.float64))
            if std is not None:
                tiff_io.imwrite(base + std_suffix, std.astype(np.float64))
        else:
            for c in range(val.shape[-1]):
                name = gs.CH_STR.get(c, str(c))
                tiff_io.imwrite(base + acq_suffix.replace(".tif", f" {name}.tif"), val[:, :, c])
                if std is not None:
                    tiff_io.imwrite(base + std_suffix.replace(".tif", f" {name}.tif"), std[:, :, c])

    def save_8bit(self, save_path: Optional[Path] = None, force_8_bit: Optional[bool] = False):
        """image_set.py:320-363: value image scaled to its maximum if that exceeds 1, rounded to uint8; the std image is
        written as float64 unless force_8_bit."""
        file_path = self.path.parent.joinpath("8bit", self.path.name) if save_path is None else Path(save_path)
        file_path.parent.mkdir(parents=True, exist_ok=True)
        val, std = self.host_arrays()
        val = val.astype(np.float64, copy=True)
        max_float = np.amax(val)
        if max_float > 1:
            val /= max_float
        tiff_io.imwrite(file_path, np.around(val * gs.MAX_DN).astype(np.uint8))
        if std is not None:
            std = std.astype(np.float64, copy=True)
            if force_8_bit:
                max_float = np.amax(std)
                if max_float > 1:
                    std /= max_float
                std = np.around(std * gs.MAX_DN).astype(np.uint8)
            tiff_io.imwrite(str(file_path).removesuffix(".tif") + " STD.tif", std)

    def save_npy(self, save_path: Path, is_HDR: bool = False):
        """Host-side dump of val (and std) as .npy next to each other ('<name> HDR.npy', '<name> HDR STD.npy' -
        the naming of save_64bit, image_set.py:285-290)."""
        val, std = self.host_arrays()
        base = str(save_path).removesuffix(Path(save_path).suffix)
        np.save(base + (" HDR" if is_HDR else "") + ".npy", val)
        if std is not None:
            np.save(base + (" HDR STD" if is_HDR else " STD") + ".npy", std)

    @staticmethod
    def compute_difference(short_exposure_set: "ImageSet", long_exposure_set: "ImageSet"):
        """image_set.py:437-451."""
        ratio = short_exposure_set.features["exposure"] / long_exposure_set.features["exposure"]
        a, r = HipMeasurand.compute_difference(short_exposure_set.measurand, long_exposure_set.measurand, ratio)
        return (ImageSet(file_path=short_exposure_set.path, features=short_exposure_set.features, measurand=a),
                ImageSet(file_path=short_exposure_set.path, features=short_exposure_set.features, measurand=r))

    @staticmethod
    def exposure_interpolation(short_exposure_set: "ImageSet", long_exposure_set: "ImageSet", exp: float):
        """image_set.py:453-480."""
        if not isinstance(exp, float):
            raise TypeError("Interpolation point has unsupported type.")
        exp0 = short_exposure_set.features["exposure"]
        exp1 = long_exposure_set.features["exposure"]
        if exp > exp1 or exp < exp0:
            raise ValueError("Interpolation point is not between the reference values.")
        m = HipMeasurand.interpolate(short_exposure_set.measurand, long_exposure_set.measurand, exp0, exp1, exp)
        return ImageSet(features=short_exposure_set.features, measurand=m)

    @classmethod
    def multiple_from_path(cls, path: Path, use_cupy: Optional[bool] = False):
        """image_set.py:482-501 (also picks up .npy files)."""
        out = []
        root = path if hasattr(path, "glob") else Path(path)
        for pattern in ("*.tif", "*.npy"):
            for file in sorted(root.glob(pattern)):
                if "STD" not in file.name:
                    out.append(cls(file_path=file, use_cupy=use_cupy))
        return out


def _bias_subtract_directory(bias_dir: Path, source_dir: Path, save_dir: Path):
    """Common body of calibrate_flats / calibrate_dark_frames (image_set.py:504-539): the shortest-exposure frame of `bias_dir`
    is the bias; every image of `source_dir` has it subtracted (Measurand.__sub__, with uncertainty) and is saved as 8-bit
    under `save_dir` with its own file name."""
    darks = ImageSet.multiple_from_path(Path(bias_dir))
    if not darks:
        raise FileNotFoundError(f"no dark frames in {bias_dir}")
    darks.sort(key=lambda image_set: image_set.features["exposure"])
    bias = darks[0]
    bias.load_value_image()
    bias.load_std_image()
    out = []
    for image_set in ImageSet.multiple_from_path(Path(source_dir)):
        image_set.load_value_image()
        image_set.load_std_image()
        image_set.measurand = image_set.measurand - bias.measurand
        image_set.save_8bit(Path(save_dir) / image_set.path.name)
        out.append(image_set)
    return out


def calibrate_flats(dark_path: Path, uncalibrated_flat_path: Path, flat_path: Path):
    """image_set.py:504-521 with the three directories as arguments (the reference reads them from its config:
    gs.DEFAULT_DARK_PATH, gs.UNCALIBRATED_FLAT_PATH, gs.DEFAULT_FLAT_PATH)."""
    return _bias_subtract_directory(dark_path, uncalibrated_flat_path, flat_path)


def calibrate_dark_frames(uncalibrated_dark_path: Path, dark_path: Path):
    """image_set.py:524-539: bias-subtract the raw dark frames (the bias is the shortest of them) into `dark_path`."""
    return _bias_subtract_directory(uncalibrated_dark_path, uncalibrated_dark_path, dark_path)


def _features_from_file_name(file_path: Path):
    """image_set.py:542-568."""
    feature_dict = {"illumination": "", "magnification": "", "exposure": 0.0, "subject": ""}
    name = file_path.name
    for suf in (".tif", ".npy"):
        name = name.removesuffix(suf)
    for element in name.split():
        if element.casefold() == "bf" or element.casefold() == "df":
            feature_dict["illumination"] = element
        elif re.match("^[0-9]+.*[xX]$", element):
            feature_dict["magnification"] = element
        elif re.match("^[0-9]+.*ms$", element):
            feature_dict["exposure"] = float(element.removesuffix("ms")) / 1000
        else:
            feature_dict["subject"] = element
    return feature_dict
