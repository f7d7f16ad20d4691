#!/usr/bin/env python3
"""Variant sweep of the val-only fused merge (config 2: 7 x 4096 x 4096 x 3 uint8) on one MI355X.

Needs a library built with -DHM_TUNE_NF=7 (the default of csrc/Makefile). Every variant is first
checked against the default variant's output, then timed with HIP events on torch's current stream
(the stream the kernels are launched on). Writes gpurun_out/tune_merge.json.

variant = 1000*TAB + 100*PREFETCH + 10*U + BLOCK_CODE   (see hm_merge.hip)
"""
import argparse
import json
import sys
import pathlib
import time

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

TAB_NAMES = {0: "plain", 1: "fused", 2: "rep16", 3: "rep32w", 4: "fused8", 5: "NONE(probe)"}


def time_sustained(plan, iters=200, warm=20):
    """average over `iters` back-to-back launches (what bench.py measures): one event pair around all."""
    for _ in range(warm):
        plan.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        plan.launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def time_plan(plan, iters, warm=3):
    for _ in range(warm):
        plan.launch()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        plan.launch()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)   # us
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=7)
    ap.add_argument("--h", type=int, default=4096)
    ap.add_argument("--w", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--random-dn", action="store_true", help="uniform random DNs instead of the radiance stack")
    ap.add_argument("--out", default="gpurun_out/tune_merge.json")
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    frames, _, t = synthetic_stack_device(7, a.n, a.h, a.w, device=dev, uniform_dn=a.random_dn)
    icrf, diff = synthetic_icrf()
    ref_plan = engine.plan_merge(frames, t, icrf, diff, variant=0)
    ref_plan.launch()
    torch.cuda.synchronize()
    ref = ref_plan.outputs["val"].clone()
    nbytes = ref_plan.algorithmic_bytes
    if a.variants:
        variants = [int(v) for v in a.variants.split(",")]
    else:
        variants = [0] + [1000 * tab + 100 * pf + 10 * u + bc for tab in (0, 1, 2, 4, 5) for pf in (0, 1)
                          for u in (1, 2, 4) for bc in (0, 1)]
    rows = []
    for v in variants:
        try:
            plan = engine.plan_merge(frames, t, icrf, diff, variant=v)
            plan.launch()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            print(f"variant {v}: {e}")
            continue
        ok = True
        if v // 1000 != 5:
            ok = bool(torch.equal(plan.outputs["val"], ref))
        med, best = time_plan(plan, a.iters)
        sus = time_sustained(plan)
        row = dict(variant=v, tab=TAB_NAMES.get(v // 1000, "?") if v else "default", prefetch=(v // 100) % 10, u=(v // 10) % 10,
                   block=1024 if v % 10 else 256, median_us=round(med, 1), min_us=round(best, 1), sustained_us=round(sus, 1),
                   GBps=round(nbytes / med / 1e3, 1), frac_8TBps=round(nbytes / med / 1e3 / 8000, 4), equal_to_default=ok)
        rows.append(row)
        print(row, flush=True)
        del plan
    rows.sort(key=lambda r: r["sustained_us"])
    pathlib.Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    json.dump(dict(config=dict(n=a.n, h=a.h, w=a.w, random_dn=a.random_dn, algorithmic_bytes=nbytes), rows=rows),
              open(a.out, "w"), indent=1)
    print("best:", rows[:5])


if __name__ == "__main__":
    main()
