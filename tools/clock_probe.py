#!/usr/bin/env python3
"""Shader clock the chip holds under the merge (and idle): hm_debug_clock_probe on a side stream while config-2 merges run on the main
stream; beside it the launch time of the merge, of round 1's kernel and (tuning / probe builds) of the table-free traffic probe."""
import json
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
icrf, _ = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, 7, 4096, 4096, device=dev)
plans = {"val3": engine.plan_merge(frames, t, icrf), "fast(r1)": engine.plan_merge(frames, t, icrf, variant=1120)}
try:
    plans["probe"] = engine.plan_merge(frames, t, icrf, variant=5120)
    plans["probe"].launch()
except Exception:
    plans.pop("probe", None)
side = torch.cuda.Stream(dev)
buf = torch.zeros(2, dtype=torch.int64, device=dev)


def clock_during(plan, ms=60):
    n = max(1, int(ms * 1e-3 / 137e-6)) if plan is not None else 0
    spins = int(ms * 0.6e-3 * 2.0e9 / (127 * 64))             # ~60 % of the window at ~2 GHz: s_sleep 127 = 127 * 64 cycles
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        plan.launch()
        if i == n // 5:
            nat.check(nat.lib.hm_debug_clock_probe(buf.data_ptr(), spins, side.cuda_stream), "probe")
    if n == 0:
        nat.check(nat.lib.hm_debug_clock_probe(buf.data_ptr(), spins, side.cuda_stream), "probe")
    e1.record()
    torch.cuda.synchronize()
    cyc, ticks = [int(x) for x in buf.cpu()]
    return round(cyc / ticks / 10, 3), (round(e0.elapsed_time(e1) * 1e3 / n, 2) if n else None)


for _ in range(3000):
    plans["val3"].launch()
torch.cuda.synchronize()
out = {}
for name, plan in plans.items():
    res = [clock_during(plan) for _ in range(3)]
    out[name] = {"clock_GHz": [r[0] for r in res], "launch_us": [r[1] for r in res]}
time.sleep(0.5)
out["idle"] = {"clock_GHz": [clock_during(None)[0] for _ in range(2)]}
print(json.dumps(out))
