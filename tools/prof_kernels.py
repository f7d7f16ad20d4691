#!/usr/bin/env python3
"""A handful of launches of k_welford and k_energy_partial for rocprofv3 --pmc passes (no pre-warm loop: counter collection
serialises every dispatch). usage: prof_kernels.py"""
import pathlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
H, W = 4096, 4096
g = torch.Generator(device=dev).manual_seed(1)
clip = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(32)]
mean = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
m2 = torch.zeros_like(mean)
for _ in range(4):
    engine.welford_update(clip, 5, mean, m2)
torch.cuda.synchronize()
del clip, mean, m2
rng = np.random.default_rng(2)
X = Y = 1024
N, B = 7, 75
t = 1e-3 * 2.0 ** np.arange(N)
dn = torch.as_tensor(rng.integers(0, 256, (X, Y, N), dtype=np.uint8), device=dev)
sd = torch.as_tensor(0.004 * (1 + rng.random((X, Y, N))), device=dev)
icrfs = np.linspace(0, 1, 256)[None, :] ** np.linspace(1.2, 3.0, B)[:, None]
icrfs[:, -1] = 1.0
icrfs_d = torch.as_tensor(icrfs, device=dev)
for _ in range(3):
    engine.linearity_energy(dn, None, t, icrfs_d, 5, 250)
    engine.linearity_energy(dn, sd, t, icrfs_d, 5, 250)
torch.cuda.synchronize()
print("done")
