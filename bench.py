#!/usr/bin/env python3
"""bench.py - HDR-merged Mpix/s and fraction of the HBM roofline on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one fused merge launch over one synthetic exposure stack that is already resident in HBM.
Workload (BASELINE.json configs[1], the one the metric is quoted on): 7 x 4096 x 4096 x 3 uint8 frames
+ 256-entry ICRF, val-only merge -> float64 radiance. At N > 1 every rank merges its own stack of the
same size (independent units, no data-path collective; "weak" scaling); value = all ranks' pixels /
max-over-ranks time. The roofline block prices the dominant kernel (merge_u8_fast) from its average
launch duration measured with HIP events on the launch stream inside the timed region; the cpu_baseline
block times the NumPy oracle (a port of the reference's merge arithmetic) on a bounded row band of the
same stack on the host, rank 0, N = 1 only.
"""
import argparse
import json
import os
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (frames, H, W, with_std, corrections)
    "cfg2": (7, 4096, 4096, False, False),      # BASELINE.json configs[1] - the headline
    "cfg3": (7, 4096, 4096, True, True),        # configs[2]: + float64 std, dark hot-pixel maps, flat field
    "cfg3std": (7, 4096, 4096, True, False),    # configs[2] without the corrections (std propagation only)
    "cfg3flat": (7, 4096, 4096, True, "flat"),  # std + flat field
    "cfg3hot": (7, 4096, 4096, True, "hot"),    # std + dark hot-pixel maps
    "cfg4tile": (15, 1024, 8192, False, False),  # configs[3]: one of 8 row tiles of 15 x 8192 x 8192 x 3
    "cfg4tilestd": (15, 1024, 8192, True, False),  # configs[3] "+std" variant of the same tile
    "cfg5": (7, 4096, 4096, False, False),      # configs[4]: 64 independent config-2 stacks on 8 GPUs = 8 resident stacks per GPU, one step merges all 8
    "cfg2rand": (7, 4096, 4096, False, False),  # config 2 with uniform-random DNs: the worst case for LDS bank conflicts
    "cfg2f64": (7, 4096, 4096, False, False),   # 64-bit mode (image_set.py:225): float64 frames, analytic weights, computed index
    "cfg3f64std": (7, 4096, 4096, True, False),  # 64-bit mode with std
}


def cpu_baseline(frames, t, icrf, diff, stds, band_rows):
    """NumPy oracle on rows [0, band_rows) of the bench stack, single thread (NumPy elementwise ops do not
    multi-thread). kind = "port": the oracle restates the reference's arithmetic (oracle/hdr_oracle.py)."""
    from oracle import hdr_oracle as orc
    fh = [f[:band_rows].cpu().numpy() for f in frames]
    sh = None if stds is None else [s[:band_rows].cpu().numpy() for s in stds]
    t0 = time.perf_counter()
    out = orc.merge(fh, t, icrf, diff, stds=sh)
    dt = time.perf_counter() - t0
    return out, dt


def cpu_baseline_threaded(frames, t, icrf, diff, stds, rows, threads):
    """The same oracle, row-tiled over a thread pool (NumPy releases the GIL inside its element-wise loops): the all-core
    number SURVEY.md 8(d) asks to show beside the single-thread one. Bands of 128 rows."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import hdr_oracle as orc
    fh = [f[:rows].cpu().numpy() for f in frames]
    sh = None if stds is None else [s[:rows].cpu().numpy() for s in stds]
    bands = [(r, min(r + 128, rows)) for r in range(0, rows, 128)]

    def one(b):
        r0, r1 = b
        orc.merge([f[r0:r1] for f in fh], t, icrf, diff, stds=None if sh is None else [s[r0:r1] for s in sh])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(one, bands))
    return time.perf_counter() - t0


def producer_workload(a, dev):
    """The SURVEY.md 8(f) rows as bench workloads (one GPU): same timing discipline as the merge (pre-warm, warm-up, K
    launches between two events) and a cpu_baseline leg = the oracle on a bounded sample. Prints one JSON line."""
    from camera_linearity_amd import engine, _native as nat
    steps, warmup = min(a.steps, 50), min(a.warmup, 5)
    if a.workload == "welford":
        H, W, K = 4096, 4096, 32
        g = torch.Generator(device=dev).manual_seed(1)
        clip = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(K)]
        mean = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
        m2 = torch.zeros_like(mean)
        launch = lambda: engine.welford_update(clip, 5, mean, m2)                       # noqa: E731
        alg = int(nat.lib.hm_welford_algorithmic_bytes(K, 1, H * W * 3))
        units, unit, metric = H * W * K, "Mpix-frames/s", "Welford-folded Mpix-frames/s (mean + M2, 32 frames per launch)"
        cfg = {"workload": f"{K} uint8 4096x4096x3 frames folded into float64 mean / M2 per launch", "name": "welford"}
        kernel, bound_note = "k_welford", "FP64 VALU from ~12 frames per launch up (DESIGN.md 4.5); frac is against the HBM roofline"
    else:
        X = Y = 1024
        N, B = 7, 75
        rng = np.random.default_rng(2)
        t = 1e-3 * 2.0 ** np.arange(N)
        dn_h = rng.integers(0, 256, (X, Y, N), dtype=np.uint8)
        dn = torch.as_tensor(dn_h, device=dev)
        icrfs = np.linspace(0, 1, 256)[None, :] ** np.linspace(1.2, 3.0, B)[:, None]
        icrfs[:, -1] = 1.0
        icrfs_d = torch.as_tensor(icrfs, device=dev)
        launch = lambda: engine.linearity_energy(dn, None, t, icrfs_d, 5, 250)          # noqa: E731
        alg = X * Y * N * B                                                              # stack bytes x candidates
        units, unit, metric = B, "candidates/s", "ICRF-calibration energy evaluations/s (1024x1024x7 stack, 75 candidates per launch)"
        cfg = {"workload": "energy function of 75 candidate ICRFs on a 1024x1024x7 uint8 channel stack per launch", "name": "energy"}
        kernel, bound_note = "k_energy_pixel", "FP64 VALU (DESIGN.md 4.5); frac is against the HBM roofline and not the relevant bound"
    t_end = time.perf_counter() + a.prewarm_s
    while time.perf_counter() < t_end:
        launch()
        torch.cuda.synchronize()
    for _ in range(warmup):
        launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        launch()
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / steps
    cpu = None
    if not a.no_cpu_baseline:
        from oracle import hdr_oracle as orc
        if a.workload == "welford":
            sample = [f[:2048].cpu().numpy() for f in clip]
            c0 = time.perf_counter()
            orc.welford_state(sample, None, True)
            dt = time.perf_counter() - c0
            cpu = {"value": round(2048 * W * K / dt / 1e6, 3), "unit": unit, "cores": 1, "kind": "port",
                   "sample": f"{K} frames of 2048x{W}x3 ({dt:.1f} s), NumPy oracle, 1 thread"}
        else:
            reps = 10
            c0 = time.perf_counter()
            for b in range(reps):
                orc.energy_function(icrfs[b], dn_h, None, 5, 250, t)
            dt = time.perf_counter() - c0
            cpu = {"value": round(reps / dt, 4), "unit": unit, "cores": 1, "kind": "port",
                   "sample": f"{reps} of the 75 candidates on the full 1024x1024x7 stack ({dt:.1f} s), NumPy oracle, 1 thread"}
    scale_u = 1e6 if unit.startswith("M") else 1.0
    line = {"metric": metric, "value": round(steps * units / elapsed / scale_u, 2), "unit": unit, "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": cfg,
            "roofline": {"bound": "hbm", "achieved": round(alg / avg_us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg / avg_us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": kernel,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(avg_us, 2), "note": bound_note},
            "cpu_baseline": cpu}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + ["welford", "energy"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise the process group even at world size 1")
    ap.add_argument("--prewarm-s", type=float, default=0.5, help="seconds of untimed launches before the warm-up steps")
    ap.add_argument("--cpu-rows", type=int, default=4096, help="rows of the stack the CPU baseline merges")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                       # --force-dist without a launcher: a one-rank group
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k_, v_)
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=a.dist_backend)
        # the first collective creates the RCCL communicator (hundreds of ms during which the GPU idles and drops its
        # clock): pay for it here, not inside the barrier that opens the timed region
        dist.barrier()
        torch.cuda.synchronize()

    if a.workload in ("welford", "energy"):          # single-GPU workloads with their own line
        producer_workload(a, dev)
        return
    from camera_linearity_amd import engine
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark

    n, H, W, with_std, corr = WORKLOADS[a.workload]
    frames, stds, t = synthetic_stack_device(7 + rank, n, H, W, device=dev, with_std=with_std)
    if "f64" in a.workload:
        frames = [engine.u8_to_unit(f) for f in frames]
    if a.workload == "cfg2rand":
        gen = torch.Generator(device=dev).manual_seed(7 + rank)
        frames = [torch.randint(0, 256, f.shape, dtype=torch.uint8, device=dev, generator=gen) for f in frames]
    icrf, diff = synthetic_icrf()
    kw = {}
    if corr:
        flat, flat_std, dark = synthetic_flat_dark(7 + rank, H, W, device=dev)
        x0, x1, y0, y1 = engine.flat_roi_bounds(H, W, 0.2)
        if corr in (True, "flat"):
            kw.update(flat=flat, flat_std=flat_std, ff_mean=engine.roi_mean(flat, x0, x1, y0, y1).cpu().numpy(),
                      ff_std_mean=engine.roi_mean(flat_std, x0, x1, y0, y1).cpu().numpy())
        if corr in (True, "hot"):
            kw.update(darks=[dark] * n, dark_min=[engine.dark_min_dn(1.0, 0.05)] * n, median_k=3)
    plan = engine.plan_merge(frames, t, icrf, diff if with_std else None, stds, variant=a.variant, **kw)
    alg_bytes = plan.algorithmic_bytes
    stacks_per_step = 1
    if a.workload == "cfg5":                 # 8 distinct resident stacks per GPU; a step = 8 launches (one per stack)
        stacks_per_step = 8
        plans = [plan]
        for k in range(1, stacks_per_step):
            fk, _, _ = synthetic_stack_device(100 * k + 7 + rank, n, H, W, device=dev, with_std=False)
            plans.append(engine.plan_merge(fk, t, icrf, None, None, variant=a.variant))

        class _Batch:                        # same interface as a MergePlan for the timing code below
            outputs = plan.outputs

            @staticmethod
            def launch():
                for p_ in plans:
                    p_.launch()
        plan = _Batch

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock pre-warm (untimed, like data generation): an idle MI355X needs a few hundred ms of load before it
    # holds its sustained clock; without it the first launches of a short run measure the ramp, not the kernel
    t_end = time.perf_counter() + a.prewarm_s
    while time.perf_counter() < t_end:
        for _ in range(50):
            plan.launch()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        plan.launch()
    # HIP events on the launch stream (torch's current stream is the one hm_merge is given), bracketing the
    # K back-to-back launches of the timed region: average launch duration = event span / K. (An event pair
    # around EVERY launch would put a barrier packet between consecutive kernels and measure a different,
    # serialised pipeline.) A second pass after the timed region records per-launch events for min/max only.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        plan.launch()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / a.steps / stacks_per_step      # per hm_merge launch
    per = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(a.steps, 20))]
    for e0, e1 in per:
        e0.record()
        plan.launch()
        e1.record()
    torch.cuda.synchronize()
    kernel_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in per]
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # parity spot check of the timed configuration against the oracle on a row band (not timed)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not corr:
        rows = min(a.cpu_rows, H)
        ref, dt = cpu_baseline(frames, t, icrf, diff, stds, rows)
        got = plan.outputs["val"][:rows].cpu().numpy()
        err = np.abs(got - ref["val"])
        denom = np.abs(ref["val"])
        nz = denom > 0                                   # radiance is exactly 0 where every frame has DN 0 (ICRF[0] = 0)
        max_rel = float(max(np.max(err[nz] / denom[nz]) if nz.any() else 0.0, np.max(err[~nz]) if (~nz).any() else 0.0))
        cpu = {"value": round(rows * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
               "sample": f"rows 0..{rows - 1} of the bench stack ({n}x{rows}x{W}x3, {dt:.1f} s), NumPy oracle, 1 thread of "
                         f"{os.cpu_count()} host cores",
               "gpu_vs_oracle_max_rel_err": max_rel, "parity_ok": bool(max_rel <= 1e-12)}
        threads = min(16, os.cpu_count() or 1)          # the one-GPU box's CPU share
        if threads > 1:
            dt_t = cpu_baseline_threaded(frames, t, icrf, diff, stds, rows, threads)
            cpu["threaded"] = {"value": round(rows * W / dt_t / 1e6, 4), "unit": "Mpix/s", "cores": threads,
                               "sample": f"same rows, 128-row bands on a {threads}-thread pool ({dt_t:.1f} s)"}

    if rank == 0:
        achieved = alg_bytes / avg_us / 1e3          # GB/s
        traffic = None
        tp = ROOT / "profiles" / "r01_pmc_traffic.json"
        if tp.exists() and a.workload == "cfg2":
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        mpix = world * a.steps * stacks_per_step * H * W / elapsed / 1e6
        line = {
            "metric": "HDR-merged Mpix/s (node)", "value": round(mpix, 1), "unit": "Mpix/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}x{H}x{W}x3 uint8 exposure stack per GPU, 256-entry ICRF LUT, "
                                   + ("float64 std + dark hot-pixel maps + flat field" if corr else
                                      ("float64 std propagation" if with_std else "val-only merge"))
                                   + " -> float64 radiance" + (" + uncertainty" if with_std else ""),
                       "name": a.workload, "frames": n, "height": H, "width": W, "channels": 3,
                       "parallelism": f"independent stacks x{world * stacks_per_step} ({stacks_per_step} resident per GPU), no collective", "variant": a.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "merge_u8_fast", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": round(avg_us, 2), "isolated_launch_us_min_median": [round(float(np.min(kernel_us)), 2), round(float(np.median(kernel_us)), 2)]},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
