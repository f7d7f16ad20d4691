#!/usr/bin/env python3
"""channel / pair statistics on one 4096 x 4096 x 3 frame, timed like tools/bench_ops.py (a subset of it: A/B runs of hm_stats.hip builds)."""
import pathlib
import statistics
import sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
shape = (4096, 4096, 3)
E = 4096 * 4096 * 3
v = torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
v2 = torch.rand(shape, dtype=torch.float64, device=dev, generator=g) + 0.5
sd = 0.01 + 0.01 * torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
sd2 = 0.01 + 0.01 * torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
cases = {"channel_statistics weighted": (lambda: engine.channel_statistics(v, sd), 16 * E), "channel_statistics unweighted": (lambda: engine.channel_statistics(v, None), 8 * E),
         "pair_statistics weighted": (lambda: engine.pair_statistics(v, sd, v2, sd2, 0.5), 32 * E), "pair_statistics unweighted": (lambda: engine.pair_statistics(v, None, v2, None, 0.5), 16 * E)}
for name, (fn, nbytes) in cases.items():
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 40)
    us = statistics.median(ts)
    print(f"{name:32s} {us:8.1f} us  {nbytes / us / 8e6:.3f} of 8 TB/s", flush=True)
