#!/usr/bin/env python3
"""PCIe-inclusive rate of config 2 (never the bench `value`): pinned host uint8 frames -> device, fused merge,
float64 result -> pinned host. Reported in DESIGN.md section 8."""
import json
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
n, H, W = 7, 4096, 4096
frames_d, _, t = synthetic_stack_device(7, n, H, W, device=dev)
host_frames = [f.cpu().pin_memory() for f in frames_d]
out_host = torch.empty((H, W, 3), dtype=torch.float64).pin_memory()
icrf, diff = synthetic_icrf()
dst = [torch.empty_like(f) for f in frames_d]
plan = engine.plan_merge(dst, t, icrf, diff)
res = {}
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d, h in zip(dst, host_frames):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    plan.launch()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out_host.copy_(plan.outputs["val"], non_blocking=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    res = dict(h2d_ms=(t1 - t0) * 1e3, merge_ms=(t2 - t1) * 1e3, d2h_ms=(t3 - t2) * 1e3, total_ms=(t3 - t0) * 1e3,
               h2d_GBps=n * H * W * 3 / (t1 - t0) / 1e9, d2h_GBps=H * W * 3 * 8 / (t3 - t2) / 1e9,
               mpix_per_s_pcie_inclusive=H * W / (t3 - t0) / 1e6)
print(json.dumps({k: round(v, 3) for k, v in res.items()}))
