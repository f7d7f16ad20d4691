"""ExposureSeries / ExposurePair - mirror of modules/exposure_series.py:18-495 on the HIP backend.

The reference's merge is two Python loops over the frames (exposure_series.py:334-341 and :372-392),
each iteration a couple of dozen whole-image NumPy temporaries plus disk reads. Here
`process_HDR_image` hands the whole stack to ONE fused HIP launch (engine.plan_merge -> hm_merge):
sum of weights, linearization, weighted radiance, variance propagation, hot-pixel filtering and the
flat-field step happen per pixel in registers, every input byte is read once.

Deviations: C/D - accumulators are plain zero-initialised arrays and S, S**2 are arrays;
E - the caller passes ICRF (and optionally ICRF_diff, else it is derived with the reference's
gradient convention); G - the corrections are applied; J - frames are taken from memory (they are
loaded from `path` only when an ImageSet holds no pixels yet).
"""
from __future__ import annotations

import math

from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import settings as gs
from .image_set import ImageSet


def icrf_derivative(ICRF, bits: int = None):
    """ICRF_diff[:, c] = np.gradient(ICRF[:, c], 2 / (BITS - 1)) - the convention of
    modules/general_functions.py:268-272 and tests/unit/test_measurand.py:21 (host, 256 x C numbers)."""
    bits = gs.BITS if bits is None else bits
    icrf = ICRF.cpu().numpy() if isinstance(ICRF, torch.Tensor) else np.asarray(ICRF, dtype=np.float64)
    dx = 2 / (bits - 1)
    if icrf.ndim == 1:
        return np.gradient(icrf, dx)
    out = np.zeros_like(icrf)
    for c in range(icrf.shape[1]):
        out[:, c] = np.gradient(icrf[:, c], dx)
    return out


def _roi_means(flat_set, eng, t: torch.Tensor, bounds, which: str) -> np.ndarray:
    """Per-channel ROI means of a flat field (modules/measurand.py:570-583) as host numbers - remembered on the flat's ImageSet: one flat
    serves many exposure series, and each evaluation is a small reduction kernel plus a device-to-host copy that the merge would wait
    for. The key holds the tensor's identity and version (an in-place edit or a new image computes again)."""
    cache = flat_set.__dict__.setdefault("_roi_mean_cache", {})       # which ("val" / "std") -> (key, means): one image and one ROI each
    key = (t.data_ptr(), t._version, tuple(t.shape), str(t.device), tuple(bounds))
    if cache.get(which, (None, None))[0] != key:
        cache[which] = (key, eng.roi_mean(t, *bounds).cpu().numpy())
    return cache[which][1]


def read_ICRF_file(file_path, return_derivative: bool = True):
    """modules/general_functions.py:254-277 with the intended return value (the derivative, not the ICRF twice)."""
    icrf = np.loadtxt(file_path, dtype=float)
    return (icrf, icrf_derivative(icrf)) if return_derivative else (icrf, None)


class ExposurePair(object):
    """modules/exposure_series.py:18-76."""

    def __init__(self, short_exposure: ImageSet, long_exposure: ImageSet):
        self.short_exposure = short_exposure
        self.long_exposure = long_exposure
        self.exposure_ratio = short_exposure.features["exposure"] / long_exposure.features["exposure"]
        self.absolute_difference = None
        self.relative_difference = None
        self.absolute_stats = None
        self.relative_stats = None

    def compute_difference(self):
        self.absolute_difference, self.relative_difference = ImageSet.compute_difference(self.short_exposure, self.long_exposure)

    def compute_stats(self, axis=None, release_memory_after: Optional[bool] = True):
        self.absolute_stats = self.absolute_difference.measurand.compute_dimension_statistics(axis=axis)
        self.relative_stats = self.relative_difference.measurand.compute_dimension_statistics(axis=axis)
        if release_memory_after:
            self.absolute_difference = None
            self.relative_difference = None

    def compute_difference_stats(self):
        """compute_difference() + compute_stats(axis=(0, 1), release_memory_after=True) in one fused HIP reduction
        (hm_pair_statistics): the two difference images are never written to HBM."""
        xs, ys = self.short_exposure.measurand, self.long_exposure.measurand
        ab, rel = xs._eng().pair_statistics(xs._f64(), xs._std, ys._f64(), ys._std, self.exposure_ratio)
        self.absolute_stats = {k: xs._export(v) for k, v in ab.items()}
        self.relative_stats = {k: xs._export(v) for k, v in rel.items()}
        self.absolute_difference = None
        self.relative_difference = None

    def process_linearity_distribution(self, bins: int, included_range=None, channels=None, use_std: Optional[bool] = False):
        return (self.absolute_difference.measurand.compute_channel_histogram(bins, included_range, channels, use_std),
                self.relative_difference.measurand.compute_channel_histogram(bins, included_range, channels, use_std))


class ExposureSeries(object):
    """modules/exposure_series.py:79-476."""

    def __init__(self, merged_image_set: Optional[ImageSet] = None, directory_path: Optional[Path] = None,
                 input_image_sets: Optional[List[ImageSet]] = None, use_cupy: Optional[bool] = True):
        self.merged_image_set = merged_image_set
        self.input_image_sets = input_image_sets if input_image_sets is not None else []
        if isinstance(directory_path, Path) and directory_path.suffix != "":
            self.directory_path = directory_path.parent
        else:
            self.directory_path = directory_path
        self.exposure_pairs = None
        self._use_cupy = use_cupy if not input_image_sets else input_image_sets[0].use_cupy

    @property
    def use_cupy(self):
        return self._use_cupy

    @use_cupy.setter
    def use_cupy(self, new_value):
        raise AttributeError("use_cupy is a read-only attribute, managing the state of the used array backend.")

    # ---- constructors (exposure_series.py:117-203)
    @classmethod
    def from_image_set(cls, reference_image_set: ImageSet, directory_path: Optional[Path] = None):
        search_path = reference_image_set.path.parent if directory_path is None else directory_path
        found = [s for s in ImageSet.multiple_from_path(search_path, use_cupy=reference_image_set.use_cupy) if reference_image_set.is_exposure_match(s)]
        found.sort(key=lambda s: s.features["exposure"])
        return cls(directory_path=search_path, input_image_sets=found)

    @classmethod
    def from_dir_path(cls, directory_path: Path, use_cupy: Optional[bool] = False):
        """exposure_series.py:148-160; `use_cupy` (an addition) picks the backend of the ImageSets it creates - the reference's
        ImageSet default, the host backend, unless asked for the device."""
        return cls.from_multiple_image_sets(ImageSet.multiple_from_path(directory_path, use_cupy=use_cupy))

    @classmethod
    def from_multiple_image_sets(cls, list_of_image_sets: List[ImageSet]):
        sublists: List[List[ImageSet]] = []
        for image_set in list_of_image_sets:
            for sub in sublists:
                if sub[0].is_exposure_match(image_set):
                    sub.append(image_set)
                    break
            else:
                sublists.append([image_set])
        out = []
        for sub in sublists:
            sub.sort(key=lambda s: s.features["exposure"])
            out.append(cls(input_image_sets=sub))
        return out

    def load_value_images(self, bit_64: Optional[bool] = False):
        for image_set in self.input_image_sets:
            image_set.load_value_image(bit64=bit_64)

    def load_std_images(self, bit_64: Optional[bool] = False):
        for image_set in self.input_image_sets:
            image_set.load_std_image(bit64=bit_64)

    def linearize(self, ICRF, ICRF_diff=None, release_memory: Optional[bool] = False):
        """exposure_series.py:226-250."""
        new_sets = []
        for s in self.input_image_sets or []:
            new_sets.append(s.linearize(ICRF, ICRF_diff))
            if release_memory:
                s.measurand.val = None
                s.measurand.std = None
        return ExposureSeries(merged_image_set=self.merged_image_set, directory_path=self.directory_path, input_image_sets=new_sets)

    def extract(self, channels=None, release_memory: Optional[bool] = False):
        new_merged = self.merged_image_set.extract(channels) if self.merged_image_set is not None else None
        new_sets = []
        for s in self.input_image_sets or []:
            new_sets.append(s.extract(channels))
            if release_memory:
                s.measurand.val = None
                s.measurand.std = None
        return ExposureSeries(merged_image_set=new_merged, directory_path=self.directory_path, input_image_sets=new_sets)

    def initialize_exposure_pairs(self):
        """exposure_series.py:283-304."""
        pairs = []
        for i, x in enumerate(self.input_image_sets):
            for j, y in enumerate(self.input_image_sets):
                if i >= j or x.features["exposure"] / y.features["exposure"] < 0.1:
                    continue
                pairs.append(ExposurePair(x, y))
        self.exposure_pairs = pairs

    # ---- the merge (exposure_series.py:317-419)
    def _eng(self):
        """The engine of this series' backend (the first image's: HIP library, or the host build inside nat.host_mode())."""
        return self.input_image_sets[0].measurand._eng()

    def _stack_inputs(self, list_of_dark_fields, dark_threshold, with_std):
        """Collect the tensors for the fused launch: frames (all uint8 DNs or all float64 values), stds, per-frame dark DN maps + DN
        thresholds - in HBM on the HIP backend, host tensors on the host backend. A dark frame held by the other backend is moved over."""
        from . import engine
        sets = self.input_image_sets
        for s in sets:
            if s.measurand.shape is None:
                s.load_value_image()
        all_dn = all(s.measurand._dn_t() is not None for s in sets)
        frames = [s.measurand._dn_t() if all_dn else s.measurand._f64() for s in sets]
        dev = frames[0].device
        stds = None
        if with_std:
            for s in sets:
                if s.measurand._std is None:
                    s.load_std_image()
                if s.measurand._std is None:
                    raise ValueError("uncertainty propagation needs a std image for every frame")
            stds = [s.measurand._std for s in sets]
        thr = gs.DARK_THRESHOLD if dark_threshold is None else dark_threshold
        darks, mins = None, None
        if list_of_dark_fields:
            darks, mins = [], []
            for s in sets:
                dark, scale = s.select_dark_field(list_of_dark_fields)    # exposure gate: gs.DARK_THRESHOLD (image_set.py:173); `thr` is the PIXEL threshold (:387)
                if dark is None:
                    darks.append(None)
                    mins.append(gs.BITS)
                    continue
                if dark.measurand.shape is None:
                    dark.load_value_image()
                if dark.measurand._dn_t() is not None:
                    darks.append(dark.measurand._dn_t().to(dev))
                    mins.append(engine.dark_min_dn(scale, thr))
                else:                                   # float-valued dark: 0/1 map, hot iff value*scale > thr
                    darks.append(((dark.measurand._tv() * scale) > thr).to(torch.uint8).to(dev))
                    mins.append(1)
        return frames, stds, darks, mins

    def _precalculate_sum_of_weights(self, list_of_dark_fields: Optional[List[ImageSet]] = None,
                                     dark_threshold: Optional[float] = None):
        """exposure_series.py:317-345 -> (S, S**2) device arrays."""
        frames, _, darks, mins = self._stack_inputs(list_of_dark_fields, dark_threshold, with_std=False)
        exp = self.input_image_sets[0].measurand._export
        S, S2 = self._eng().sum_of_weights(frames, darks=darks, dark_min=mins, median_k=gs.MEDIAN_FILTER_KERNEL_SIZE)
        return exp(S), exp(S2)

    def _compute_HDR_image_set(self, list_of_dark_fields, sum_of_weights, square_sum_of_weights, ICRF, ICRF_diff,
                               flat_set: Optional[ImageSet] = None, dark_threshold: Optional[float] = None,
                               use_std: Optional[bool] = None):
        """exposure_series.py:347-397. The fused kernel recomputes the sum of weights in registers, so the
        two precalculated arrays are accepted for signature compatibility and not read."""
        from . import engine
        eng = self._eng()
        sets = self.input_image_sets
        if use_std is None:
            for s in sets:                                  # the reference loads every frame's std image (:377),
                if s.measurand._std is None and s.path is not None:     # whether or not its value image is in memory yet
                    s.load_std_image()
            have = [s.measurand._std is not None for s in sets]
            if any(have) and not all(have):
                raise ValueError("some frames have a std image and some do not: uncertainty propagation needs one for every frame "
                                 "(pass use_std=False to merge values only)")
            use_std = all(have)
        frames, stds, darks, mins = self._stack_inputs(list_of_dark_fields, dark_threshold, with_std=use_std)
        dev = frames[0].device
        if ICRF_diff is None and use_std:
            ICRF_diff = icrf_derivative(ICRF)
        kw = {}
        if flat_set is not None:                                   # exposure_series.py:415-417
            if flat_set.measurand.shape is None:
                flat_set.load_value_image()
            fm = flat_set.measurand
            fval = (fm._dn_t() if fm._dn_t() is not None else fm._f64()).to(dev)
            size_x = gs.IM_SIZE_X or fval.shape[0]
            size_y = gs.IM_SIZE_Y or fval.shape[1]
            x0, x1, y0, y1 = engine.flat_roi_bounds(size_x, size_y, gs.FF_MID_PERCENTAGE)
            kw.update(flat=fval, ff_mean=_roi_means(flat_set, eng, fval, (x0, x1, y0, y1), "val"))
            if use_std:
                if fm._std is None:
                    flat_set.load_std_image()
                if fm._std is None:
                    raise ValueError("flat field needs a std image to propagate uncertainty")
                fstd = fm._std.to(dev)
                kw.update(flat_std=fstd, ff_std_mean=_roi_means(flat_set, eng, fstd, (x0, x1, y0, y1), "std"))
        exposures = [s.features["exposure"] for s in sets]
        out = eng.merge(frames, exposures, ICRF, ICRF_diff if use_std else None, stds, darks=darks, dark_min=mins,
                        median_k=gs.MEDIAN_FILTER_KERNEL_SIZE, **kw)
        hdr = type(sets[0].measurand)(out["val"], out.get("std"))
        hdr_set = ImageSet(file_path=sets[0].get_file_path_without_exposure(), features=dict(sets[0].features) if sets[0].features else None,
                           measurand=hdr)
        hdr_set.is_HDR = True
        return hdr_set

    def process_HDR_image(self, ICRF=None, ICRF_diff=None, dark_list: Optional[List[ImageSet]] = None,
                          flat_list: Optional[List[ImageSet]] = None, use_std: Optional[bool] = None):
        """exposure_series.py:399-419: merge the input images into self.merged_image_set.

        Defaults mirror the reference: `ICRF=None` reads settings.ICRF_CALIBRATED_FILE (:406-407; deviation E - the table is
        read as ONE (BITS, C) array and the derivative is formed with the reference's gradient convention), `dark_list=None`
        globs settings.DEFAULT_DARK_PATH (:409) and `flat_list=None` settings.DEFAULT_FLAT_PATH (image_set.py:146-155). A
        path that is not configured means "none": no dark frames / no flat field - but a missing ICRF raises."""
        if ICRF is None:
            if gs.ICRF_CALIBRATED_FILE is None:
                raise ValueError("process_HDR_image(ICRF=None) needs settings.ICRF_CALIBRATED_FILE (settings.configure(...)) "
                                 "or an explicit ICRF array")
            ICRF, default_diff = read_ICRF_file(gs.ICRF_CALIBRATED_FILE)
            if ICRF_diff is None:
                ICRF_diff = default_diff
        if not self.input_image_sets:
            raise ValueError("no input images")
        if dark_list is None and gs.DEFAULT_DARK_PATH is not None:
            dark_list = ImageSet.multiple_from_path(Path(gs.DEFAULT_DARK_PATH), use_cupy=self.use_cupy)
        if flat_list is None and gs.DEFAULT_FLAT_PATH is not None:
            flat_list = ImageSet.multiple_from_path(Path(gs.DEFAULT_FLAT_PATH), use_cupy=self.use_cupy)
        flat_set = self.input_image_sets[0].get_flat_field(flat_list) if flat_list else None
        self.merged_image_set = self._compute_HDR_image_set(dark_list, None, None, ICRF, ICRF_diff, flat_set=flat_set,
                                                            use_std=use_std)

    # ---- linearity statistics (exposure_series.py:421-476; "next" row f-1)
    def process_linearity(self, ICRF, linearity_limit: Optional[int] = None, use_std: Optional[bool] = False):
        lower, upper = map_linearity_limits(linearity_limit, linearity_limit, ICRF)
        for image_set in self.input_image_sets:
            if image_set.measurand.shape is None:
                image_set.load_value_image()
            if image_set.measurand._std is None and use_std:
                image_set.load_std_image()
        # the thresholds ride on the all-pairs launch where that applies (no separate pass over the frames); otherwise frame by frame
        if self._all_pairs_fused(lower, upper):
            return
        for image_set in self.input_image_sets:
            image_set.measurand.apply_thresholds(lower, upper)
        if self._all_pairs_fused():
            return
        for pair in self.exposure_pairs:
            v = pair.short_exposure.measurand._tv()
            if v is not None and v.dim() == 3 and v.shape[-1] <= 4 and pair.long_exposure.measurand.shape == tuple(v.shape):
                pair.compute_difference_stats()              # fused: no difference images in HBM
            else:
                pair.compute_difference()
                pair.compute_stats(axis=(0, 1), release_memory_after=True)

    def _all_pairs_fused(self, lower=None, upper=None) -> bool:
        """Every pair of the series in ONE launch (hm_pairs_statistics): each frame is read from HBM once instead of once per
        pair it takes part in. Applies when the pairs are pairs of this series' own images, all images share one (H, W, C <= 4)
        shape on one device and either all or none of them carry a std; otherwise the per-pair path above runs."""
        sets = self.input_image_sets
        if not self.exposure_pairs or not sets:
            return False
        index = {id(s): i for i, s in enumerate(sets)}
        if any(id(p.short_exposure) not in index or id(p.long_exposure) not in index for p in self.exposure_pairs):
            return False
        shapes = {s.measurand.shape for s in sets}
        if len(shapes) != 1 or None in shapes or len(next(iter(shapes))) != 3 or next(iter(shapes))[-1] > 4:
            return False
        have_std = [s.measurand._std is not None for s in sets]
        if any(have_std) != all(have_std):
            return False
        vals = [s.measurand._f64() for s in sets]
        if not all(v.device == vals[0].device for v in vals) or len({s.measurand.backend for s in sets}) != 1:
            return False
        stds = [s.measurand._std for s in sets] if all(have_std) else None
        pairs = [(index[id(p.short_exposure)], index[id(p.long_exposure)], p.exposure_ratio) for p in self.exposure_pairs]
        thresholds = None
        if lower is not None or upper is not None:           # apply_thresholds (measurand.py:375-428) inside the launch, in place
            n_ch = vals[0].shape[-1]
            lower = [None] * n_ch if lower is None else lower
            upper = [None] * n_ch if upper is None else upper
            if len(lower) != n_ch or len(upper) != n_ch:
                raise ValueError("The length of 'lower' and 'upper' must match the size of the independent axis.")
            thresholds = ([-math.inf if l is None else l for l in lower], [math.inf if u is None else u for u in upper])
            vals = [v.contiguous() for v in vals]
            if stds is not None:
                stds = [sd.to(torch.float64).contiguous() for sd in stds]
            for s_, v in zip(sets, vals):                      # the image sets keep the tensors that are thresholded in place
                s_.measurand.val = v
            if stds is not None:
                for s_, sd in zip(sets, stds):
                    s_.measurand.std = sd
        exp = sets[0].measurand._export
        for p, (ab, rel) in zip(self.exposure_pairs, self._eng().pairs_statistics(vals, stds, pairs, to_host=True, thresholds=thresholds)):
            # (host tensors: 6C numbers per pair, fetched with ONE copy; NumPy arrays on the host backend)
            p.absolute_stats, p.relative_stats = {k: exp(v) for k, v in ab.items()}, {k: exp(v) for k, v in rel.items()}
            p.absolute_difference = p.relative_difference = None
        return True

    def collect_exposure_pair_stats(self, return_cupy: Optional[bool] = False):
        rel = {"ratios": [], "means": [], "stds": [], "errors": []}
        ab = {"ratios": [], "means": [], "stds": [], "errors": []}
        for p in self.exposure_pairs:
            for res, st in ((ab, p.absolute_stats), (rel, p.relative_stats)):
                res["ratios"].append(p.exposure_ratio)
                res["means"].append(_host(st["mean"]))
                res["stds"].append(_host(st["std"]))
                res["errors"].append(_host(st["error"]))
        return _to_2d_array(ab), _to_2d_array(rel)


def _host(x):
    return None if x is None else (x.cpu().numpy() if isinstance(x, torch.Tensor) else x)


def _to_2d_array(dictionary: Dict):
    return {k: np.array(v) for k, v in dictionary.items()}


def map_linearity_limits(lower_limit: Optional[int], upper_limit: Optional[int], ICRF):
    """modules/general_functions.py:97-131 (host; returns per-channel lists)."""
    n = gs.NUM_OF_CHS
    lower = np.array([gs.LOWER_LIN_LIM if lower_limit is None else lower_limit] * n, dtype="float64")
    upper = np.array([gs.UPPER_LIN_LIM if upper_limit is None else gs.MAX_DN - upper_limit] * n, dtype="float64")
    if ICRF is None:
        lower /= gs.MAX_DN
        upper /= gs.MAX_DN
    else:
        icrf = ICRF.cpu().numpy() if isinstance(ICRF, torch.Tensor) else np.asarray(ICRF)
        for c in range(n):
            lower[c] = icrf[int(lower[c]), c]
            upper[c] = icrf[int(upper[c]), c]
    return list(lower), list(upper)
