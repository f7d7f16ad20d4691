#!/usr/bin/env python3
"""Wall time of ExposureSeries.process_HDR_image through the Python API on device-resident image sets (a camera-sized stack:
7 x 2048 x 2448 x 3 uint8 + float64 std, flat field), against the kernel time of the same merge (events around hm_merge alone)."""
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.exposure_series import ExposureSeries  # noqa: E402
from camera_linearity_amd.image_set import ImageSet  # noqa: E402
from camera_linearity_amd.measurand import HipMeasurand  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark  # noqa: E402

dev = torch.device("cuda:0")
n, H, W = 7, 2048, 2448
frames, stds, t = synthetic_stack_device(7, n, H, W, device=dev, with_std=True)
icrf, diff = synthetic_icrf()
flat, flat_std, dark = synthetic_flat_dark(7, H, W, device=dev)
feat = lambda ti, subj="x": {"exposure": float(ti), "illumination": "bf", "magnification": "5x", "subject": subj}   # noqa: E731
for use_std in (False, True):
    sets = [ImageSet(measurand=HipMeasurand(f, s if use_std else None), features=feat(ti)) for f, s, ti in zip(frames, stds, t)]
    flats = [ImageSet(measurand=HipMeasurand(flat, flat_std if use_std else None), features=feat(0.01, "flat"))]
    for with_flat in (False, True):
        series = ExposureSeries(input_image_sets=sets)
        for _ in range(3):
            series.process_HDR_image(icrf, diff if use_std else None, flat_list=flats if with_flat else None)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            series.process_HDR_image(icrf, diff if use_std else None, flat_list=flats if with_flat else None)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        plan = engine.plan_merge(frames, t, icrf, diff if use_std else None, stds if use_std else None)
        plan.launch(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.launch()
        e1.record(); torch.cuda.synchronize()
        print(f"use_std={use_std} flat={with_flat}: process_HDR_image {wall * 1e6:.0f} us per call; merge kernel alone (no flat) {e0.elapsed_time(e1) * 1e3 / reps:.0f} us", flush=True)
