"""Synthetic exposure stacks of SURVEY.md section 8(d), generated ON THE DEVICE (the bench needs
7 x 50 MB frames; building them with NumPy on the host would dominate its run time).

The recipe is the one the parity tests use on the host (oracle.hdr_oracle.synthetic_stack): a random
radiance map `rad` in [0, 4), exposures t_i = 1 ms * 2**i, frame_i = clip(around(rad * t_i * k), 0, 255)
with k = 255 / (4 t_{N//2}) - short frames mostly dark, long frames mostly saturated. Only torch's
RNG differs from NumPy's, so device stacks are not bit-equal to host stacks of the same seed; parity
tests that need identical inputs copy the device stack back.
"""
from __future__ import annotations

import numpy as np
import torch


def synthetic_exposures(n: int) -> np.ndarray:
    return 1e-3 * 2.0 ** np.arange(n)


def synthetic_stack_device(seed: int, n: int, h: int, w: int, c: int = 3, device="cuda", with_std: bool = False,
                           uniform_dn: bool = False, smooth: bool = False):
    """-> (frames [n x (h,w,c) uint8], stds [n x (h,w,c) float64] | None, exposures ndarray).
    smooth: the radiance is a low-frequency pattern + 1 % noise instead of independent per element, so that saturated and
    under-exposed pixels form regions as in a photograph (thresholded frames are then NaN in whole areas)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    t = synthetic_exposures(n)
    frames = []
    if uniform_dn:
        for _ in range(n):
            frames.append(torch.randint(0, 256, (h, w, c), generator=gen, device=device, dtype=torch.uint8))
    else:
        rad = torch.rand((h, w, c), generator=gen, device=device, dtype=torch.float64) * 4
        if smooth:
            yy = torch.arange(h, device=device, dtype=torch.float64).view(h, 1, 1) / h
            xx = torch.arange(w, device=device, dtype=torch.float64).view(1, w, 1) / w
            cc = 1.0 + 0.1 * torch.arange(c, device=device, dtype=torch.float64).view(1, 1, c)
            rad = (2.0 + 1.9 * torch.sin(6.283185307179586 * 3 * xx) * torch.cos(6.283185307179586 * 2 * yy)) * cc / 1.2 + 0.01 * rad
            rad = torch.clamp(rad, 0.0, 4.0)
        k = 255 / (4 * t[n // 2])
        for ti in t:
            frames.append(torch.clamp(torch.round(rad * float(ti * k)), 0, 255).to(torch.uint8))
        del rad
    stds = None
    if with_std:
        stds = [0.004 * (1 + torch.rand((h, w, c), generator=gen, device=device, dtype=torch.float64)) for _ in range(n)]
    return frames, stds, t


def synthetic_icrf(gammas=(2.2, 2.0, 1.8), bits: int = 256):
    """ICRF = linspace(0,1,bits)**gamma_c; derivative with the reference's convention
    dx = 2/(BITS-1) (modules/general_functions.py:270, tests/unit/test_measurand.py:21)."""
    icrf = np.stack([np.linspace(0, 1, bits) ** g for g in gammas], axis=1)
    diff = np.stack([np.gradient(icrf[:, c], 2 / (bits - 1)) for c in range(len(gammas))], axis=1)
    return icrf, diff


def synthetic_flat_dark(seed: int, h: int, w: int, c: int = 3, device="cuda", hot_density: float = 1e-4):
    """flat = clip(around(255 (0.8 + 0.05 u))) uint8 with std 0.002; dark = dim background with sparse hot pixels."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1000)
    flat = torch.clamp(torch.round(255 * (0.8 + 0.05 * torch.rand((h, w, c), generator=gen, device=device, dtype=torch.float64))),
                       0, 255).to(torch.uint8)
    flat_std = torch.full((h, w, c), 0.002, dtype=torch.float64, device=device)
    dark = torch.randint(0, 4, (h, w, c), generator=gen, device=device, dtype=torch.uint8)
    hot = torch.rand((h, w, c), generator=gen, device=device) < hot_density
    dark[hot] = 200
    return flat, flat_std, dark
