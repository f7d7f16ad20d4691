#!/usr/bin/env python3
"""Throughput of the SURVEY.md 8(f) rows next to the merge (GPU side only; the CPU baselines of these rows are the
`cpu_baseline` legs of `bench.py --workload welford` / `--workload energy`):
  - Welford mean / M2 producer (hm_welford_update): 4096 x 4096 x 3 frames, K frames per launch, against its algorithmic
    bytes  E * (K + 32)  [+16 without M2];
  - ICRF-calibration energy function (hm_linearity_energy): one DE generation (75 candidates) per launch at the
    reference's default stack size (data_spacing=150 -> 28 x 28 x 7) and at 1024 x 1024 x 7.
Pre-warmed; median of rounds. Writes gpurun_out/bench_producers.json."""
import json
import pathlib
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402

dev = torch.device("cuda:0")
out = {"welford": [], "energy": []}


def span(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def prewarm():
    x = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    t0 = time.time()
    while time.time() - t0 < 0.5:
        x.add_(1)
    torch.cuda.synchronize()


QUICK = "--quick" in sys.argv          # profiling runs: fewer repetitions


def median_us(fn, iters=20, rounds=5):
    if QUICK:
        iters, rounds = 5, 1
    prewarm()
    return statistics.median(span(fn, iters) for _ in range(rounds))


# ---------------------------------------------------------------- Welford
H, W = 4096, 4096
E = H * W * 3
g = torch.Generator(device=dev).manual_seed(1)
clip = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(32)]
mean = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
m2 = torch.zeros_like(mean)
icrf = np.linspace(0, 1, 256)[:, None] ** np.array([2.2, 2.0, 1.8])[None, :]
for K, with_m2, lut in ((8, True, None), (16, True, None), (32, True, None), (32, False, None), (32, True, icrf)):
    fr = clip[:K]
    us = median_us(lambda: engine.welford_update(fr, 5, mean, m2 if with_m2 else None, lut))
    b = nat.lib.hm_welford_algorithmic_bytes(K, int(with_m2), E)
    row = {"frames_per_launch": K, "m2": with_m2, "icrf": lut is not None, "us": round(us, 1), "GB/s": round(b / us / 1e3, 1),
           "frac_of_8TB/s": round(b / us / 1e3 / 8000, 3), "Mpix_frames/s": round(H * W * K / us, 1)}
    out["welford"].append(row)
    print("welford", row, flush=True)
del clip, mean, m2

# ---------------------------------------------------------------- energy function
rng = np.random.default_rng(2)
for (X, Y, N, B) in ((28, 28, 7, 75), (256, 256, 7, 75), (1024, 1024, 7, 75), (1024, 1024, 7, 1)):
    t = 1e-3 * 2.0 ** np.arange(N)
    dn_h = rng.integers(0, 256, (X, Y, N), dtype=np.uint8)
    sd_h = 0.004 * (1 + rng.random((X, Y, N)))
    dn, sd = torch.as_tensor(dn_h, device=dev), torch.as_tensor(sd_h, device=dev)
    gam = np.linspace(1.2, 3.0, B)
    icrfs = np.linspace(0, 1, 256)[None, :] ** gam[:, None]
    icrfs[:, -1] = 1.0
    icrfs_d = torch.as_tensor(icrfs, device=dev)
    for with_std in (False, True):
        us = median_us(lambda: engine.linearity_energy(dn, sd if with_std else None, t, icrfs_d, 5, 250), iters=10, rounds=3)
        pair_px = X * Y * (N * (N - 1) // 2) * B
        row = {"stack": f"{X}x{Y}x{N}", "candidates": B, "std": with_std, "us_per_launch": round(us, 1),
               "us_per_candidate": round(us / B, 2), "Gpair_px/s": round(pair_px / us / 1e3, 2)}
        out["energy"].append(row)
        print("energy", row, flush=True)

pathlib.Path("gpurun_out").mkdir(exist_ok=True)
pathlib.Path("gpurun_out/bench_producers.json").write_text(json.dumps(out, indent=1))
