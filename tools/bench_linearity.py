#!/usr/bin/env python3
"""ExposureSeries.process_linearity end to end on a device-resident series: 7 linearized 4096 x 4096 x 3
float64 frames (+ std), all exposure pairs with ratio >= 0.1. Prints the time per call and per pair."""
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd.exposure_series import ExposureSeries  # noqa: E402
from camera_linearity_amd.image_set import ImageSet  # noqa: E402
from camera_linearity_amd.measurand import HipMeasurand  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402
from camera_linearity_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
n, H, W = 7, 4096, 4096
SMOOTH = "--smooth" in sys.argv      # photograph-like radiance: the thresholds leave NaN REGIONS instead of isolated NaNs
frames, stds, t = synthetic_stack_device(7, n, H, W, device=dev, with_std=True, smooth=SMOOTH)
icrf, diff = synthetic_icrf()
for use_std in (False, True):
    sets = []
    for f, s, ti in zip(frames, stds, t):
        val, sd = engine.linearize(f, s if use_std else None, icrf, diff if use_std else None)
        sets.append(ImageSet(measurand=HipMeasurand(val, sd), features={"exposure": float(ti), "illumination": "bf", "magnification": "5x", "subject": "x"}))
    series = ExposureSeries(input_image_sets=sets)
    series.initialize_exposure_pairs()
    npairs = len(series.exposure_pairs)
    for rep in range(3):
        # thresholds are applied in place: re-linearize cheaply is not needed for timing (NaNs stay NaNs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        series.process_linearity(icrf, linearity_limit=5, use_std=use_std)
        ab, rel = series.collect_exposure_pair_stats()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    E = H * W * 3
    # the kernels alone (device-resident thresholded frames): all pairs in one launch vs one fused launch per pair
    vals = [s_.measurand._f64() for s_ in sets]
    sds = [s_.measurand.std for s_ in sets] if use_std else None
    idx = {id(s_): i for i, s_ in enumerate(sets)}
    pairs = [(idx[id(p.short_exposure)], idx[id(p.long_exposure)], p.exposure_ratio) for p in series.exposure_pairs]

    def timed(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_all = timed(lambda: engine.pairs_statistics(vals, sds, pairs))
    from camera_linearity_amd.exposure_series import map_linearity_limits
    lo_, hi_ = map_linearity_limits(5, 5, icrf)
    t_thr = timed(lambda: engine.pairs_statistics(vals, sds, pairs, thresholds=(lo_, hi_)))
    t_sep = timed(lambda: [engine.apply_thresholds_(v_, None if sds is None else sds[i_], lo_, hi_) for i_, v_ in enumerate(vals)])
    t_each = timed(lambda: [engine.pair_statistics(vals[i], None if sds is None else sds[i], vals[j], None if sds is None else sds[j], m) for i, j, m in pairs])
    once = n * (2 if use_std else 1) * 8 * E                            # every frame (+ std) read once
    nan_frac = sum(float(torch.isnan(v_).double().mean()) for v_ in vals) / len(vals)
    print(f"smooth={SMOOTH} NaN fraction {nan_frac:.2f} use_std={use_std}: {npairs} pairs; process_linearity end to end {dt * 1e3:.1f} ms; hm_pairs_statistics alone {t_all:.2f} ms, with the thresholds fused {t_thr:.2f} ms ({len(vals)} x hm_apply_thresholds alone: {t_sep:.2f} ms) "
          f"({once / t_all / 1e9:.2f} TB/s of read-once traffic); {npairs} x hm_pair_statistics {t_each:.2f} ms", flush=True)
    del sets, series
