#!/usr/bin/env python3
"""A few launches of the analytic Gaussian-weight kernel (hm_gaussian_weight_f64, measurand.py:606-618) on a 4096 x 4096 x 3 float64 frame
for the rocprofv3 passes of tools/profile.sh (PROG=tools/prof_weight.py)."""
import pathlib
import sys
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
dev = torch.device("cuda:0")
v = torch.rand((4096, 4096, 3), dtype=torch.float64, device=dev)
for _ in range(40):
    engine.gaussian_weight(v)
torch.cuda.synchronize()
print("done")
