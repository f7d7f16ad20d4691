#!/usr/bin/env python3
"""Does the placement of the frames in HBM matter? The 7 frames of a config-2 stack are 48 MiB each; every wave reads the same offset of all
seven at the same time, so if the DRAM address mapping does not hash strides of 48 MiB apart, the seven streams compete for the same
channels / banks. Times the val-only merge with the frames (a) in separate torch allocations, (b-...) as views of one buffer at
stride 48 MiB + pad for several pads. Interleaved rounds on one box; prints one JSON line."""
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
n, H, W = 7, 4096, 4096
E = H * W * 3
icrf, _ = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, n, H, W, device=dev)
PADS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else [0, 4096])]


def placed(pad):
    buf = torch.empty(n * (E + pad) + 4096, dtype=torch.uint8, device=dev)
    out = []
    for i, f in enumerate(frames):
        v = buf[i * (E + pad): i * (E + pad) + E].view(H, W, 3)
        v.copy_(f)
        out.append(v)
    return buf, out


cases = {"separate allocations": (None, frames)}
for pad in PADS:
    cases[f"one buffer, stride 48 MiB + {pad}"] = placed(pad)
plans = {k: engine.plan_merge(v[1], t, icrf) for k, v in cases.items()}
ref = None
for k, p in plans.items():
    p.launch()
    torch.cuda.synchronize()
    ref = p.outputs["val"] if ref is None else ref
    assert torch.equal(p.outputs["val"], ref), k


def span(p, iters=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        p.launch()
    e0.record()
    for _ in range(iters):
        p.launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for _ in range(3000):
    plans["separate allocations"].launch()
torch.cuda.synchronize()
res = {k: [] for k in plans}
for _ in range(7):
    for k, p in plans.items():
        res[k].append(span(p))
out = {k: round(statistics.median(v), 2) for k, v in res.items()}
out["frame_ptrs_mod_2MiB"] = [f.data_ptr() % (2 << 20) for f in frames]
out["frame_ptr_deltas"] = [frames[i + 1].data_ptr() - frames[i].data_ptr() for i in range(n - 1)]
print(json.dumps(out))
