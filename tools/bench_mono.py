#!/usr/bin/env python3
"""Monochrome stacks (C = 1): 7 x 4096 x 4096 x 1 and 7 x 6144 x 8192 x 1 (the element count of config 2), val-only and with std."""
import pathlib, statistics, sys
import numpy as np
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
dev = torch.device("cuda:0")
n = 7
g = torch.Generator(device=dev).manual_seed(5)
icrf = np.linspace(0, 1, 256)[:, None] ** 2.2
diff = np.gradient(icrf, 2 / 255, axis=0)
t = list(1e-3 * 2.0 ** np.arange(n))
for H, W in ((4096, 4096), (6144, 8192)):
    rad = torch.rand((H, W, 1), generator=g, device=dev, dtype=torch.float64) * 4
    k = 255 / (4 * t[n // 2])
    frames = [torch.clamp(torch.round(rad * float(ti * k)), 0, 255).to(torch.uint8) for ti in t]
    stds = [0.004 * (1 + torch.rand((H, W, 1), generator=g, device=dev, dtype=torch.float64)) for _ in range(n)]
    for with_std in (False, True):
        plan = engine.plan_merge(frames, t, icrf, diff if with_std else None, stds if with_std else None)
        for _ in range(200):
            plan.launch()
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                plan.launch()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 50)
        us = statistics.median(ts)
        print(f"{n}x{H}x{W}x1 std={with_std}: {us:8.1f} us  {plan.algorithmic_bytes / us / 8e6:.3f} of 8 TB/s  {plan.kernels}", flush=True)
    del frames, stds, rad
