#!/usr/bin/env python3
"""Placement experiment 2: frames AND the output live in one allocation at controlled relative offsets; every configuration is built
twice (two separate allocations) to see whether the time is a function of the relative layout or of where the allocation landed."""
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
n, H, W = 7, 4096, 4096
E = H * W * 3
icrf, _ = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, n, H, W, device=dev)
CONFIGS = [tuple(int(x) for x in c.split(":")) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0:0", "4096:0", "16384:0", "20480:8192"])]


def build(pad, opad):
    total = n * (E + pad) + opad + 8 * E + (4 << 20)
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    fr = []
    for i, f in enumerate(frames):
        v = buf[i * (E + pad): i * (E + pad) + E].view(H, W, 3)
        v.copy_(f)
        fr.append(v)
    o0 = n * (E + pad) + opad
    o0 = (o0 + 15) // 16 * 16
    out = buf[o0:o0 + 8 * E].view(torch.float64).view(H, W, 3)
    plan = engine.plan_merge(fr, t, icrf)
    plan.args.out_val = out.data_ptr()                     # redirect the output into the same allocation
    plan.outputs["val"] = out
    return buf, plan


plans = {}
for pad, opad in CONFIGS:
    for rep in ("A", "B"):
        plans[f"pad {pad} out+{opad} {rep}"] = build(pad, opad)
ref = None
for k, (b, p) in plans.items():
    p.launch()
    torch.cuda.synchronize()
    ref = p.outputs["val"].clone() if ref is None else ref
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


first = next(iter(plans.values()))[1]
for _ in range(3000):
    first.launch()
torch.cuda.synchronize()
res = {k: [] for k in plans}
for _ in range(7):
    for k, (b, p) in plans.items():
        res[k].append(span(p))
out = {k: [round(statistics.median(v), 2), hex(plans[k][0].data_ptr())] for k, v in res.items()}
print(json.dumps(out))
