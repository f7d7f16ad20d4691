#!/usr/bin/env python3
"""Launch time of the config-2 merge for every (input stack, output buffer) pair of S resident stacks and K output buffers, and the
4 KB-stride translation probe (hm_debug_stride_probe) of every buffer involved: which side of the kernel does the launch-time lottery of
DESIGN.md 4.4 come from, and does it follow the buffers' physical backing? Prints a table."""
import pathlib
import statistics
import sys

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import engine, _native as nat  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf  # noqa: E402

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
icrf, _ = synthetic_icrf()
stacks = [synthetic_stack_device(7 + 100 * s, 7, 4096, 4096, device=dev) for s in range(S)]
plans = [engine.plan_merge(f, t, icrf) for f, _, t in stacks]
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


def probe(t_):
    nbytes = t_.numel() * t_.element_size()
    return timed(lambda: nat.check(nat.lib.hm_debug_stride_probe(t_.data_ptr(), nbytes, 4096, 64, 1024, sink.data_ptr(), stream), "probe"), 10)


for _ in range(3000):
    plans[0].launch()
torch.cuda.synchronize()
res = {(s, k): [] for s in range(S) for k in range(K)}
for rep in range(3):
    for s in range(S):
        for k in range(K):
            plans[s].args.out_val = outs[k].data_ptr()
            res[(s, k)].append(timed(plans[s].launch, 30))
print("merge us: rows = input stack, columns = output buffer")
for s in range(S):
    print(f"stack{s}", " ".join(f"{statistics.median(res[(s, k)]):7.2f}" for k in range(K)), flush=True)
print("4 KB-stride probe us per output buffer:", " ".join(f"{probe(o):6.1f}" for o in outs))
for s in range(S):
    print(f"4 KB-stride probe us of stack{s}'s 7 frames:", " ".join(f"{probe(f):6.1f}" for f in stacks[s][0]), flush=True)
print("addresses: outs", [hex(o.data_ptr()) for o in outs])
for s in range(S):
    print(f"stack{s}", [hex(f.data_ptr()) for f in stacks[s][0]])
