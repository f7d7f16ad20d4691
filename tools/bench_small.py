#!/usr/bin/env python3
"""Config 1 (3 x 256 x 256 x 3) is launch-bound: a batch of such stacks merged by eager hm_merge launches vs the same
launches captured once into a hipGraph and replayed. Writes gpurun_out/bench_small.json."""
import json
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
icrf, diff = synthetic_icrf()
out = []
for (n, H, W, B, with_std) in ((3, 256, 256, 64, False), (3, 256, 256, 64, True), (7, 512, 512, 64, False)):
    plans = []
    for k in range(B):
        frames, stds, t = synthetic_stack_device(k, n, H, W, device=dev, with_std=with_std)
        plans.append(engine.plan_merge(frames, t, icrf, diff if with_std else None, stds))
    for p in plans:
        p.launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p in plans:
            p.launch()

    def timed(fn, reps=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    eager = timed(lambda: [p.launch() for p in plans])
    graph = timed(g.replay)
    row = {"stack": f"{n}x{H}x{W}x3", "std": with_std, "batch": B, "eager_us_per_stack": round(eager / B, 2),
           "graph_us_per_stack": round(graph / B, 2), "Mpix/s_eager": round(B * H * W / eager, 1), "Mpix/s_graph": round(B * H * W / graph, 1)}
    out.append(row)
    print(row, flush=True)
pathlib.Path("gpurun_out").mkdir(exist_ok=True)
pathlib.Path("gpurun_out/bench_small.json").write_text(json.dumps(out, indent=1))
