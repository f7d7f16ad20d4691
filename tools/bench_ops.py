#!/usr/bin/env python3
"""Throughput of the standalone kernels behind the reference's per-frame methods (SURVEY.md 8a rows 1-4, 8-10
and 8f-1) on one 4096 x 4096 x 3 frame, against their algorithmic bytes. Pre-warmed; median of rounds.
Writes gpurun_out/bench_ops.json."""
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark  # noqa: E402

dev = torch.device("cuda:0")
H, W = 4096, 4096
E = H * W * 3
frames, stds, t = synthetic_stack_device(7, 2, H, W, device=dev, with_std=True)
dn, sd = frames[1], stds[1]
sd2 = stds[0]          # the second operand's std is its OWN buffer: passing `sd` twice halves the real traffic of a 4-stream kernel
v = engine.u8_to_unit(dn)
v2 = engine.u8_to_unit(frames[0]) + 0.25
icrf, diff = synthetic_icrf()
icrf, diff = torch.as_tensor(icrf, device=dev), torch.as_tensor(diff, device=dev)     # tables resident on the device (an upload per call is not the kernel)
flat, flat_std, dark = synthetic_flat_dark(7, H, W, device=dev)
x0, x1, y0, y1 = engine.flat_roi_bounds(H, W, 0.2)
m = engine.roi_mean(flat, x0, x1, y0, y1).cpu().numpy()
s = engine.roi_mean(flat_std, x0, x1, y0, y1).cpu().numpy()

CASES = {
    # name: (callable, algorithmic bytes)
    "u8_to_unit (image_set.py:223)": (lambda: engine.u8_to_unit(dn), E * 9),
    "linearize u8 -> f64 (measurand.py:471)": (lambda: engine.linearize(dn, None, icrf), E * 9),
    "linearize u8 + std (ICRF_diff * std)": (lambda: engine.linearize(dn, sd, icrf, diff), E * 25),
    "linearize f64 input (round-half-even index)": (lambda: engine.linearize(v, None, icrf), E * 16),
    "gaussian weight u8 LUT (w, dw)": (lambda: engine.gaussian_weight(dn), E * 17),
    "gaussian weight f64 analytic (w, dw)": (lambda: engine.gaussian_weight(v), E * 24),
    "__add__ with std (measurand.py:106)": (lambda: engine.elementwise_binary(nat.HM_OP_ADD, v, sd, v2, sd2), E * 48),
    "__mul__ with std": (lambda: engine.elementwise_binary(nat.HM_OP_MUL, v, sd, v2, sd2), E * 48),
    "__truediv__ with std": (lambda: engine.elementwise_binary(nat.HM_OP_DIV, v, sd, v2, sd2), E * 48),
    "__pow__ array exponent (generic pow), no std": (lambda: engine.elementwise_binary(nat.HM_OP_POW, v2, None, torch.tensor([2.0], device=dev, dtype=torch.float64), None), E * 16),
    "__pow__ scalar 2, no std (S ** 2, exposure_series.py:343)": (lambda: engine.pow_scalar(v2, None, 2.0), E * 16),
    "__pow__ scalar 1/2, no std (exposure_series.py:394)": (lambda: engine.pow_scalar(v2, None, 0.5), E * 16),
    "__pow__ scalar 2 with std": (lambda: engine.pow_scalar(v2, sd, 2.0), E * 32),
    "__pow__ scalar 2.2 with std (general exponent)": (lambda: engine.pow_scalar(v2, sd, 2.2), E * 32),
    "normalize_by_map (flat field, with std)": (lambda: engine.normalize_by_map(v, sd, flat, flat_std, m, s), E * 41),
    "hot_pixel_filter u8 (dark map, 3x3 median)": (lambda: engine.hot_pixel_filter(dn, dark, 0.05, 3), E * 3),
    "hot_pixel_filter f64 (dark map, 3x3 median)": (lambda: engine.hot_pixel_filter(v, dark, 0.05, 3), E * 17),
    "roi_mean u8 (flat ROI 20 %)": (lambda: engine.roi_mean(flat, x0, x1, y0, y1), (x1 - x0) * (y1 - y0) * 3),
    "channel_statistics weighted (one pass)": (lambda: engine.channel_statistics(v, sd), E * 16),
    "channel_statistics unweighted (one pass)": (lambda: engine.channel_statistics(v, None), E * 8),
    "pair_statistics weighted (one pass, fused)": (lambda: engine.pair_statistics(v, sd, v2, sd2, 0.5), E * 32),
    "pair_statistics unweighted (one pass, fused)": (lambda: engine.pair_statistics(v, None, v2, None, 0.5), E * 16),
    "compute_difference with std": (lambda: engine.compute_difference(v, sd, v2, sd2, 0.5), E * 64),
}


def span(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for _ in range(300):
    engine.linearize(dn, sd, icrf, diff)
torch.cuda.synchronize()
res = {k: [] for k in CASES}
for _ in range(5):
    for k, (fn, b) in CASES.items():
        res[k].append(span(fn, 10))
out = []
for k, (fn, b) in CASES.items():
    us = statistics.median(res[k])
    out.append(dict(kernel=k, us=round(us, 1), algorithmic_MB=round(b / 1e6, 1), GBps=round(b / us / 1e3, 1), frac_8TBps=round(b / us / 8e6, 3)))
    print(out[-1])
pathlib.Path("gpurun_out").mkdir(exist_ok=True)
json.dump(out, open("gpurun_out/bench_ops.json", "w"), indent=1)
