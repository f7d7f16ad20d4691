#!/usr/bin/env python3
"""Channel / pair statistics launches for a rocprofv3 --kernel-trace --stats run (per-kernel durations of the two-launch reductions)."""
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
shape = (4096, 4096, 3)
v = torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
v2 = torch.rand(shape, dtype=torch.float64, device=dev, generator=g) + 0.5
sd = 0.01 + 0.01 * torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
sd2 = 0.01 + 0.01 * torch.rand(shape, dtype=torch.float64, device=dev, generator=g)
for _ in range(20):
    engine.channel_statistics(v, sd)
    engine.channel_statistics(v, None)
    engine.pair_statistics(v, sd, v2, sd2, 0.5)
    engine.pair_statistics(v, None, v2, None, 0.5)
torch.cuda.synchronize()
