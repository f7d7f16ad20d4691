#!/usr/bin/env python3
"""BASELINE config 5 on one GPU's share (8 independent 7 x 4096 x 4096 x 3 stacks), host memory to host memory:
serial (copy in, merge, copy out, one stack after the other) against MergePipeline (copies and kernels of consecutive
stacks overlapped on three HIP streams). The producer writes into the pinned views in place (one byte per stack changes,
so that every stack is really transferred). Writes gpurun_out/bench_pipeline.json."""
import json
import pathlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.pipeline import MergePipeline  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
n, H, W, STACKS = 7, 4096, 4096, 8
frames_d, _, t = synthetic_stack_device(7, n, H, W, device=dev)
icrf, diff = synthetic_icrf()
host_frames = [f.cpu().numpy() for f in frames_d]
out = {}

# serial reference: pinned staging, one stack at a time
pin_in = [torch.from_numpy(f).pin_memory() for f in host_frames]
pin_out = torch.empty((H, W, 3), dtype=torch.float64).pin_memory()
plan = engine.plan_merge(frames_d, t, icrf, None)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(STACKS):
        pin_in[0].numpy()[0, 0, 0] = k
        for d, h in zip(frames_d, pin_in):
            d.copy_(h, non_blocking=True)
        plan.launch()
        pin_out.copy_(plan.outputs["val"], non_blocking=True)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
out["serial"] = {"ms_per_stack": round(dt / STACKS * 1e3, 2), "Mpix/s": round(STACKS * H * W / dt / 1e6, 1)}
print("serial", out["serial"], flush=True)
serial_val = pin_out.numpy().copy()

for depth in (2, 3):
    pipe = MergePipeline(n, H, W, t, icrf, None, depth=depth)
    for slot in range(depth):                       # stage the stack once per slot; per stack only one byte changes
        fv, _ = pipe.input_views(slot)
        for dst, src in zip(fv, host_frames):
            np.copyto(dst, src)

    def fill(k, fv, sv):
        if k >= STACKS:
            return False
        fv[0][0, 0, 0] = k
        return True
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for k, val, std in pipe.run(fill):
            last = val
        dt = time.perf_counter() - t0
    assert np.array_equal(last, serial_val)
    out[f"pipeline_depth{depth}"] = {"ms_per_stack": round(dt / STACKS * 1e3, 2), "Mpix/s": round(STACKS * H * W / dt / 1e6, 1)}
    print(f"pipeline depth {depth}", out[f"pipeline_depth{depth}"], flush=True)
    del pipe
pathlib.Path("gpurun_out").mkdir(exist_ok=True)
pathlib.Path("gpurun_out/bench_pipeline.json").write_text(json.dumps(out, indent=1))
