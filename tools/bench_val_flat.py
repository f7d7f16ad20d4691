#!/usr/bin/env python3
"""val-only merge WITH a flat field (and with dark maps) on 7 x 4096 x 4096 x 3: the configurations between BASELINE configs 2 and 3."""
import pathlib, statistics, sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark  # noqa: E402
dev = torch.device("cuda:0")
import os
n, H, W = int(os.environ.get("BN", 7)), int(os.environ.get("BH", 4096)), int(os.environ.get("BW", 4096))
icrf, diff = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, n, H, W, device=dev)
flat, flat_std, dark = synthetic_flat_dark(7, H, W, device=dev)
x0, x1, y0, y1 = engine.flat_roi_bounds(H, W, 0.2)
m = engine.roi_mean(flat, x0, x1, y0, y1).cpu().numpy()
darks = [dark] + [synthetic_flat_dark(7 + 17 * i, H, W, device=dev)[2] for i in range(1, n)]
cases = {"val only": {}, "val + flat": dict(flat=flat, ff_mean=m), "val + 7 dark maps": dict(darks=darks, dark_min=[13] * n, median_k=3),
         "val + flat + 7 dark maps": dict(flat=flat, ff_mean=m, darks=darks, dark_min=[13] * n, median_k=3)}
for name, kw in cases.items():
    plan = engine.plan_merge(frames, t, icrf, **kw)
    for _ in range(300):
        plan.launch()
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            plan.launch()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 100)
    us = statistics.median(ts)
    print(f"{name:28s} {us:8.1f} us  {plan.algorithmic_bytes / us / 8e6:.3f} of 8 TB/s   {plan.kernels}", flush=True)
