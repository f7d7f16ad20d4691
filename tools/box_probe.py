#!/usr/bin/env python3
"""What kind of box is this? Plain device-to-device copy and fill bandwidth (torch), the merge kernel, and the shader clock it holds."""
import json, pathlib, sys, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
dev = torch.device("cuda:0")
def span(fn, it=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e-3
a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
frames, _, t = synthetic_stack_device(7, 7, 4096, 4096, device=dev)
icrf, _ = synthetic_icrf()
plan = engine.plan_merge(frames, t, icrf)
for _ in range(2000): plan.launch()
out = {"copy_1GiB_TBps": round(2 * (1 << 30) / span(lambda: b.copy_(a)) / 1e12, 3),
       "fill_1GiB_TBps": round((1 << 30) / span(lambda: b.fill_(1)) / 1e12, 3),
       "sum_1GiB_TBps": round((1 << 30) / span(lambda: a.view(torch.int64).sum()) / 1e12, 3),
       "merge_us": round(span(plan.launch, 100) * 1e6, 2), "device": torch.cuda.get_device_name(0)}
p = torch.cuda.get_device_properties(0)
out["props"] = {"cu": p.multi_processor_count, "mem_GiB": round(p.total_memory / 2**30, 1)}
print(json.dumps(out))
