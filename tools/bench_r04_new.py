#!/usr/bin/env python3
"""Round-4 kernels against their algorithmic bytes on one MI355X: the chunked merge (N > 32) and statistics over single axes.
Prints one JSON line per case: us per call, algorithmic GB/s, fraction of 8 TB/s."""
import json
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_icrf  # noqa: E402


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    icrf, diff = synthetic_icrf()
    g = torch.Generator(device=dev).manual_seed(1)
    out = []
    for n, h, w, with_std in ((64, 2048, 2048, False), (64, 2048, 2048, True), (40, 2048, 2048, True), (32, 2048, 2048, True), (32, 2048, 2048, False)):
        frames = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(n)]
        stds = [0.004 * (1 + torch.rand((h, w, 3), dtype=torch.float64, device=dev, generator=g)) for _ in range(n)] if with_std else None
        t = 1e-3 * 2.0 ** (np.arange(n) - n // 2)
        plan = engine.plan_merge(frames, t, icrf, diff if with_std else None, stds)
        us = timed(plan.launch)
        alg = plan.algorithmic_bytes
        out.append({"case": f"merge N={n} {h}x{w}x3" + (" + std" if with_std else ""), "kernel": plan.kernels, "us": round(us, 1),
                    "GBps": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / 8000, 4)})
        del frames, stds, plan
        torch.cuda.empty_cache()
    v = torch.rand((4096, 4096, 3), dtype=torch.float64, device=dev, generator=g)
    s = 0.01 + 0.1 * torch.rand((4096, 4096, 3), dtype=torch.float64, device=dev, generator=g)
    for axis in (0, 1, 2, (0, 1), (1, 2), (0, 2)):
        for sd in (None, s):
            us = timed(lambda: engine.axis_statistics(v, sd, axis))
            alg = v.numel() * 8 * (2 if sd is not None else 1)
            out.append({"case": f"axis_statistics axis={axis}" + (" weighted" if sd is not None else ""), "us": round(us, 1), "GBps": round(alg / us / 1e3, 1),
                        "frac": round(alg / us / 1e3 / 8000, 4)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
