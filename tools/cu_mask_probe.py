#!/usr/bin/env python3
"""Config-2 merge on streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): the merge runs at the package power
limit, so does it get faster, or no slower, with fewer CUs drawing power? Prints us per launch per mask."""
import ctypes as C
import pathlib
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
icrf, _ = synthetic_icrf()
stacks = [synthetic_stack_device(7 + 100 * s, 7, 4096, 4096, device=dev) for s in range(4)]
plans = [engine.plan_merge(f, t, icrf) for f, _, t in stacks]
for _ in range(3000):
    plans[0].launch()
torch.cuda.synchronize()


def run(mask_words, label):
    stream = C.c_void_p()
    arr = (C.c_uint32 * len(mask_words))(*mask_words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(stream), len(mask_words), arr)
    if rc != 0:
        print(label, "hipExtStreamCreateWithCUMask failed", rc)
        return
    ext = torch.cuda.ExternalStream(stream.value, device=dev)
    with torch.cuda.stream(ext):
        for k in range(200):
            plans[k % 4].launch(stream.value)
        res = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for k in range(80):
                plans[k % 4].launch(stream.value)
            e1.record(ext)
            ext.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 80)
    res.sort()
    print(f"{label}: {res[2]:.2f} us per launch (min {res[0]:.2f})", flush=True)
    hip.hipStreamDestroy(stream)


full = [0xFFFFFFFF] * 8
run(full, "all 256 CUs")
run([0x77777777] * 8, "3 of every 4 CUs (192)")
run([0x55555555] * 8, "every second CU (128)")
run([0xFFFFFFFF] * 7 + [0], "first 224 mask bits")
run([0xFEFEFEFE] * 8, "7 of every 8 CUs (224)")
run(full, "all 256 CUs again")
