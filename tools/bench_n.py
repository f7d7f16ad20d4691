#!/usr/bin/env python3
"""Merge throughput vs frame count N on 2048 x 4096 x 3 stacks (val-only and +std): which N take the templated kernel
(N <= 16) and what the generic kernel delivers beyond. Prints one line per case."""
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
icrf, diff = synthetic_icrf()
H, W = 2048, 4096
args = [x for x in sys.argv[1:] if not x.startswith("--")]
ns = [int(x) for x in args] or [4, 8, 12, 16, 17, 24, 32]
modes = (False,) if "--val-only" in sys.argv else (True,) if "--std-only" in sys.argv else (False, True)
for with_std in modes:
    for n in ns:
        frames, stds, t = synthetic_stack_device(3, n, H, W, device=dev, with_std=with_std)
        plan = engine.plan_merge(frames, t, icrf, diff if with_std else None, stds)
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(20):
                plan.launch()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            plan.launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        b = plan.algorithmic_bytes
        print(f"N={n:2d} std={int(with_std)} {us:8.1f} us  {b / us / 1e3:7.1f} GB/s  frac {b / us / 8e6:.3f}", flush=True)
        del frames, stds, plan
