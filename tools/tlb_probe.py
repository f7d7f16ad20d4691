#!/usr/bin/env python3
"""Does the launch-time lottery of the merge (128-130 us or 134-138 us depending on where a buffer landed) follow the
address-translation cost of the buffer? K candidate output buffers: merge time into each (shared inputs), then hm_debug_stride_probe on each
with page strides of 4 KB, 64 KB and 2 MB (one dword per page and thread: translation-bound, not bandwidth-bound). One JSON line per buffer."""
import json
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
icrf, _ = synthetic_icrf()
frames, _, t = synthetic_stack_device(7, 7, 4096, 4096, device=dev)
plan = engine.plan_merge(frames, t, icrf)
outs = [torch.empty((4096, 4096, 3), dtype=torch.float64, device=dev) for _ in range(K)]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def probe(buf, nbytes, stride):
    return timed(lambda: nat.check(nat.lib.hm_debug_stride_probe(buf, nbytes, stride, 64, 1024, sink.data_ptr(), stream), "probe"), 20)


for _ in range(3000):
    plan.launch()
torch.cuda.synchronize()
rows = []
for rep in range(3):
    for k, o in enumerate(outs):
        plan.args.out_val = o.data_ptr()
        m = timed(plan.launch, 40)
        nbytes = o.numel() * 8
        r = {"buf": k, "ptr": hex(o.data_ptr()), "merge_us": round(m, 2)}
        for name, stride in (("4K", 4096), ("64K", 65536), ("2M", 2 << 20)):
            r[f"probe_{name}_us"] = round(probe(o.data_ptr(), nbytes, stride), 2)
        rows.append(r)
for k in range(K):
    mine = [r for r in rows if r["buf"] == k]
    print(json.dumps({key: (mine[0][key] if key in ("buf", "ptr") else statistics.median(r[key] for r in mine)) for key in mine[0]}), flush=True)
# the input frames too (they are 50 MB each: fewer pages)
for i, f in enumerate(frames):
    print(json.dumps({"frame": i, "ptr": hex(f.data_ptr()), "probe_4K_us": round(probe(f.data_ptr(), f.numel(), 4096), 2),
                      "probe_64K_us": round(probe(f.data_ptr(), f.numel(), 65536), 2)}), flush=True)
