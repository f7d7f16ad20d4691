#!/usr/bin/env python3
"""One standalone kernel, timed like tools/bench_ops.py (pre-warmed, median of rounds): usage bench_one_op.py hot_u8|hot_f64|hot_u8_dense"""
import statistics
import sys
import pathlib
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_flat_dark  # noqa: E402
dev = torch.device("cuda:0")
H = W = 4096
E = H * W * 3
frames, _, _ = synthetic_stack_device(7, 2, H, W, device=dev)
what = sys.argv[1] if len(sys.argv) > 1 else "hot_u8"
dens = 1e-4 if "dense" not in what else 1e-2
_, _, dark = synthetic_flat_dark(7, H, W, device=dev, hot_density=dens)
x = frames[1] if "u8" in what else engine.u8_to_unit(frames[1])
nbytes = E * (3 if "u8" in what else 17)
fn = lambda: engine.hot_pixel_filter(x, dark, 0.05, 3)   # noqa: E731
for _ in range(300):
    fn()
torch.cuda.synchronize()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 50)
us = statistics.median(ts)
print(what, "density", dens, round(us, 2), "us", round(nbytes / us / 1e3, 1), "GB/s frac", round(nbytes / us / 8e6, 4))
