#!/usr/bin/env python3
"""Kernel x output-placement matrix: K candidate output buffers (separate allocations), and for each of them the launch time of
merge_u8_val3 (0), round 1's merge_u8_fast (1120) and - in -DHM_PROBE tuning builds - the table-free traffic probe (5120), all writing into
that same buffer and reading the same input stack. Separates what the kernel costs from where the buffer landed. One JSON line."""
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1120", "5120"])]
icrf, _ = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, 7, 4096, 4096, device=dev)
plans = {}
for v in variants:
    try:
        plans[v] = engine.plan_merge(frames, t, icrf, variant=v)
        plans[v].launch()
        torch.cuda.synchronize()
    except Exception as e:  # noqa
        print("variant", v, "unavailable:", e)
outs = [torch.empty((4096, 4096, 3), dtype=torch.float64, device=dev) for _ in range(K)]


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


first = next(iter(plans.values()))
for _ in range(3000):
    first.launch()
torch.cuda.synchronize()
res = {f"out{k}@{hex(o.data_ptr())}": {v: [] for v in plans} for k, o in enumerate(outs)}
for _ in range(5):
    for k, o in enumerate(outs):
        for v, p in plans.items():
            p.args.out_val = o.data_ptr()
            res[f"out{k}@{hex(o.data_ptr())}"][v].append(span(p))
table = {k: {str(v): round(statistics.median(ts), 2) for v, ts in d.items()} for k, d in res.items()}
print(json.dumps(table))
