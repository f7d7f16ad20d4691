#!/usr/bin/env python3
"""engine.PlanGraph: K small stacks (BASELINE config 1: 3 x 256 x 256 x 3, and two larger sizes) merged by K direct hm_merge calls against one
hipGraph replay. Prints host-inclusive time per stack (wall clock around enqueue + synchronize) for both."""
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
icrf, diff = synthetic_icrf()
K = 64


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for n, h, w in ((3, 256, 256), (7, 512, 512), (7, 1024, 1024), (7, 4096, 4096)):
    k = K if h < 4096 else 4
    plans = []
    for s in range(k):
        frames, _, t = synthetic_stack_device(s, n, h, w, device=dev)
        plans.append(engine.plan_merge(frames, t, icrf))
    graph = engine.PlanGraph(plans)
    wide = {lanes: engine.PlanGraph(plans, lanes=lanes) for lanes in (2, 4, 8)}

    def direct():
        for p in plans:
            p.launch()

    td = timed(direct, 20) / k
    tg = timed(graph.replay, 20) / k
    b = plans[0].algorithmic_bytes
    print(f"{n} x {h} x {w} x 3, {k} stacks: direct {td * 1e6:8.2f} us/stack ({b / td / 1e9:7.1f} GB/s)   graph replay {tg * 1e6:8.2f} us/stack "
          f"({b / tg / 1e9:7.1f} GB/s)   x{td / tg:.2f}   " +
          "  ".join(f"{lanes} lanes {timed(g.replay, 20) / k * 1e6:.2f}" for lanes, g in wide.items()), flush=True)
    del plans, graph, wide
