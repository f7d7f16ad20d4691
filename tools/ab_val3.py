#!/usr/bin/env python3
"""A/B of the val-only merge kernels on one MI355X: merge_u8_val3 (variant 0) against merge_u8_fast (variant 1120).

Checks first that both produce the same bits (and that the IEEE-division path of val3, forced by an exposure outside the
range div_inrange() is proven for, agrees with merge_generic), then times them in interleaved rounds, once re-merging ONE
resident stack and once rotating over R distinct resident stacks (inputs > the 256 MB Infinity Cache). Prints one JSON line.
"""
import argparse
import json
import pathlib
import statistics
import sys

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402


def span_us(plans, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(3):
        plans[k % len(plans)].launch()
    e0.record()
    for k in range(iters):
        plans[k % len(plans)].launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=7)
    ap.add_argument("--h", type=int, default=4096)
    ap.add_argument("--w", type=int, default=4096)
    ap.add_argument("--stacks", type=int, default=4)
    ap.add_argument("--iters", type=int, default=48)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", default="0,1120")
    ap.add_argument("--random-dn", action="store_true")
    ap.add_argument("--own-outputs", action="store_true", help="give every variant output buffers of its own (round-2 runs before the placement effect was known)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    icrf, diff = synthetic_icrf()
    variants = [int(v) for v in a.variants.split(",")]
    stacks = []
    for k in range(a.stacks):
        frames, _, t = synthetic_stack_device(7 + 100 * k, a.n, a.h, a.w, device=dev, uniform_dn=a.random_dn)
        stacks.append(frames)
    plans = {v: [engine.plan_merge(f, t, icrf, None, None, variant=v) for f in stacks] for v in variants}
    if not a.own_outputs:
        # every variant writes into the SAME output buffers (those of the first variant): launch times depend on where a buffer landed
        # in HBM by several %, so variants with outputs of their own are not comparable
        base = plans[variants[0]]
        for v in variants[1:]:
            for p_, b_ in zip(plans[v], base):
                p_.args.out_val = b_.outputs["val"].data_ptr()
                p_.outputs["val"] = b_.outputs["val"]
    nbytes = plans[variants[0]][0].algorithmic_bytes
    # ---- parity between the kernels
    ref = engine.plan_merge(stacks[0], t, icrf, None, None, variant=-1)          # merge_generic
    ref.launch()
    torch.cuda.synchronize()
    equal = {}
    for v in variants:
        plans[v][0].outputs["val"].zero_()
        plans[v][0].launch()
        torch.cuda.synchronize()
        equal[v] = bool(torch.equal(plans[v][0].outputs["val"], ref.outputs["val"]))
    t_far = list(t)
    t_far[0] = 1e-200                                                            # 1/t outside [2^-300, 2^300]: val3 must take its IEEE-division path
    slow = engine.plan_merge(stacks[0], t_far, icrf, None, None, variant=0)
    slow_ref = engine.plan_merge(stacks[0], t_far, icrf, None, None, variant=-1)
    slow.launch(); slow_ref.launch()
    torch.cuda.synchronize()
    equal["ieee_path"] = bool(torch.equal(slow.outputs["val"], slow_ref.outputs["val"]))
    del slow, slow_ref, ref
    # ---- timing
    for _ in range(4000):
        plans[variants[0]][0].launch()
    torch.cuda.synchronize()
    same = {v: [] for v in variants}
    rot = {v: [] for v in variants}
    for _ in range(a.rounds):
        for v in variants:
            same[v].append(span_us(plans[v][:1], a.iters))
        for v in variants:
            rot[v].append(span_us(plans[v], a.iters))
    res = {"config": {"n": a.n, "h": a.h, "w": a.w, "stacks": a.stacks, "random_dn": a.random_dn, "algorithmic_bytes": nbytes},
           "bit_equal_to_generic": {str(k): v for k, v in equal.items()}, "rows": []}
    for v in variants:
        ms, mr = statistics.median(same[v]), statistics.median(rot[v])
        res["rows"].append({"variant": v, "same_stack_us": round(ms, 2), "same_min": round(min(same[v]), 2), "rotating_us": round(mr, 2),
                            "rot_min": round(min(rot[v]), 2), "same_frac": round(nbytes / ms / 8e6, 4), "rotating_frac": round(nbytes / mr / 8e6, 4)})
    print(json.dumps(res))
    if a.out:
        pathlib.Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
