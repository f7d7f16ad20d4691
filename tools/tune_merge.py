#!/usr/bin/env python3
"""Variant sweep of the val-only fused merge (config 2: 7 x 4096 x 4096 x 3 uint8) on one MI355X.

Needs a library built with -DHM_TUNE_NF=7 (the default of csrc/Makefile). Method (guide rule 24): every
variant is first checked bit-for-bit against the default variant's output; the chip is pre-warmed with
~1 s of back-to-back launches; then R rounds, each timing every variant over `iters` back-to-back launches
between two HIP events on the launch stream; the median over rounds is reported.
Writes gpurun_out/tune_merge.json.      variant = 1000*TAB + 100*PREFETCH + 10*U + BLOCK_CODE (hm_merge.hip)
"""
import argparse
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

TAB_NAMES = {0: "plain", 1: "fused", 2: "rep16", 4: "fused8", 5: "NONE(probe)"}


def span_us(plan, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        plan.launch()
    e0.record()
    for _ in range(iters):
        plan.launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=7)
    ap.add_argument("--h", type=int, default=4096)
    ap.add_argument("--w", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--random-dn", action="store_true", help="uniform random DNs instead of the radiance stack")
    ap.add_argument("--out", default="gpurun_out/tune_merge.json")
    ap.add_argument("--variants", default="")
    ap.add_argument("--std", action="store_true", help="tune the std kernel (config 3 without corrections)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    frames, stds, t = synthetic_stack_device(7, a.n, a.h, a.w, device=dev, uniform_dn=a.random_dn, with_std=a.std)
    icrf, diff = synthetic_icrf()
    ref_plan = engine.plan_merge(frames, t, icrf, diff, stds, variant=0)
    ref_plan.launch()
    torch.cuda.synchronize()
    ref = ref_plan.outputs["val"]
    nbytes = ref_plan.algorithmic_bytes
    if a.variants:
        variants = [int(v) for v in a.variants.split(",")]
    elif a.std:
        variants = [0] + [100 * pf + 10 * u for pf in (0, 1) for u in (1, 2, 4)]
    else:
        variants = [0] + [1000 * tab + 100 * pf + 10 * u + bc for tab in (0, 1, 5) for pf in (0, 1) for u in (2, 4, 8) for bc in (0, 1)]
    plans = []
    shared_out = ref_plan.outputs["val"]
    for v in variants:
        try:
            plan = engine.plan_merge(frames, t, icrf, diff, stds, variant=v)
            plan.launch()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            print(f"variant {v}: {e}")
            continue
        ok = True if v // 1000 == 5 else bool(torch.equal(plan.outputs["val"], ref))
        if a.std:
            ok = ok and bool(torch.equal(plan.outputs["std"], ref_plan.outputs["std"]))
        plans.append((v, plan, ok, []))
    for _ in range(1200 if a.std else 6000):   # pre-warm to the sustained clock
        ref_plan.launch()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v, plan, ok, ts in plans:
            ts.append(span_us(plan, a.iters))
    rows = []
    for v, plan, ok, ts in plans:
        med = statistics.median(ts)
        rows.append(dict(variant=v, tab=TAB_NAMES.get(v // 1000, "?") if v else "default", prefetch=(v // 100) % 10, u=(v // 10) % 10,
                         block=1024 if v % 10 else 256, median_us=round(med, 1), min_us=round(min(ts), 1), max_us=round(max(ts), 1),
                         GBps=round(nbytes / med / 1e3, 1), frac_8TBps=round(nbytes / med / 1e3 / 8000, 4), equal_to_default=ok))
    rows.sort(key=lambda r: r["median_us"])
    for r in rows:
        print(r)
    pathlib.Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    json.dump(dict(config=dict(n=a.n, h=a.h, w=a.w, random_dn=a.random_dn, algorithmic_bytes=nbytes, rounds=a.rounds, iters=a.iters),
                   rows=rows), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
