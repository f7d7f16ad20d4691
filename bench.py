#!/usr/bin/env python3
"""bench.py - HDR-merged Mpix/s and fraction of the HBM roofline on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one fused merge launch over one synthetic exposure stack that is already resident in HBM.
Workload (BASELINE.json configs[1], the one the metric is quoted on): 7 x 4096 x 4096 x 3 uint8 frames
+ 256-entry ICRF, val-only merge -> float64 radiance. FOUR distinct stacks are resident and merged round-robin, so
no launch re-reads the inputs of the one before it (cold inputs; the figure for re-merging ONE stack is reported beside
it as roofline.same_stack_*). At N > 1 every rank merges its own stacks of the same size (independent units, no
data-path collective; "weak" scaling); value = all ranks' pixels / max-over-ranks time. The roofline block prices the
kernel hm_merge dispatches to (its name comes from the library: hm_merge_describe) from its average launch duration
measured with HIP events on the launch stream inside the timed region; the cpu_baseline block times the NumPy oracle
(a port of the reference's merge arithmetic) on a bounded row band of the same stack on the host, rank 0, N = 1 only,
and checks the GPU output of the timed configuration against it.
Other workloads (--workload): cfg3* = configs[2] and its parts, cfg4 / cfg4std = configs[3] (ONE 15 x 8192 x 8192 x 3 image
in 8 row tiles dealt to the ranks, host-side assembly), cfg5 = configs[4], cfg2rand, cfg2smooth, cfg2f64 / cfg3f64std, welford, energy, linearity / linearitystd.
"""
import argparse
import json
import os
import pathlib
import socket
import subprocess
import sys
import time

# numpy and torch are imported by main() AFTER the self-launch decision (late_imports): the parent of an N > 1 run must not
# have touched torch (let alone HIP) when it starts its ranks as a child process (tests/test_host_logic.py holds it to that).
np = None
torch = None

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def late_imports():
    global np, torch
    import numpy
    import torch as torch_
    np, torch = numpy, torch_


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(gpus, argv, runner=None):
    """`python bench.py --gpus N` (N > 1) from a plain shell: start the N ranks as a CHILD process - the same command the
    driver would type (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <same arguments>) - wait for it and return its exit code. Rank 0 of the child prints the JSON line on the
    inherited stdout. Nothing in this process has imported torch or made a HIP call at this point, and nothing is exec'ed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(ROOT / "bench.py")] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return (runner or subprocess.run)(cmd, env=env).returncode

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
    "cfg2smooth": (7, 4096, 4096, False, False),  # config 2 on a photograph-like radiance (low-frequency pattern + 1 % noise): neighbouring lanes gather neighbouring table rows
    "cfg2f64": (7, 4096, 4096, False, False),   # 64-bit mode (image_set.py:225): float64 frames, analytic weights, computed index
    "cfg3f64std": (7, 4096, 4096, True, False),  # 64-bit mode with std
}


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


def rank_report(dist, a, dev, elapsed, avg_us):
    """max-over-ranks elapsed time + what every rank measured for itself (all-gathered): the N > 1 line shows how many ranks
    took part and the spread of their average launch durations, so a straggler GPU is visible."""
    if dist is None:
        return elapsed, None
    on = dev if a.dist_backend == "nccl" else "cpu"
    mine = torch.tensor([elapsed, avg_us], dtype=torch.float64, device=on)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    el = [float(x[0]) for x in every]
    us = [float(x[1]) for x in every]
    return max(el), {"ranks_seen": len(every), "avg_launch_us_min": round(min(us), 2), "avg_launch_us_max": round(max(us), 2),
                     "avg_launch_us_per_rank": [round(x, 2) for x in us], "elapsed_s_per_rank": [round(x, 5) for x in el]}


DARK_THR = 0.05                 # pixel threshold of the dark maps (settings.DARK_THRESHOLD)


# ---- counter records (tools/profile.sh -> tools/summarize_profile.py -> profiles/r04_pmc_counters.json) --------------------------------
# A record holds what the PMC passes measured per launch of a workload's kernel (HBM bytes; VALU wave-instructions for the FP64-bound
# kernels), the kernel's name, the commit and a HASH OF THE KERNEL'S SOURCE FILES at collection time. bench.py prints a record only while
# that hash is the hash of the sources the loaded library was built from in this tree: after any edit of the kernel's file the line says
# `traffic: null, traffic_stale: true` until the profile is re-run (round 3 matched records by kernel name only).
COUNTER_RECORDS = "profiles/r04_pmc_counters.json"
_KERNEL_SOURCES = {"merge": ("hm_merge.hip", "hm_common.h"), "linearity": ("hm_stats.hip", "hm_common.h"), "welford": ("hm_welford.hip", "hm_common.h"),
                   "energy": ("hm_energy.hip", "hm_common.h")}


def kernel_source_hash(workload):
    """sha256 (first 16 hex digits) of the csrc files the workload's dominant kernel is compiled from."""
    import hashlib
    fam = "linearity" if workload.startswith("linearity") else workload if workload in ("welford", "energy") else "merge"
    h = hashlib.sha256()
    for name in _KERNEL_SOURCES[fam]:
        h.update((ROOT / "camera_linearity_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def counter_record(workload):
    """(record | None, stale): the workload's PMC record when it was collected on the current kernel sources; stale = a record exists
    but for other sources."""
    tp = ROOT / COUNTER_RECORDS
    try:
        ent = json.load(open(tp)).get(workload)
    except Exception:
        return None, False
    if not ent:
        return None, False
    if ent.get("source_hash") != kernel_source_hash(workload):
        return None, True
    return ent, False


def measured_traffic(workload, kernel=None):
    """(HBM bytes per launch | None, where it comes from | None, stale flag)."""
    ent, stale = counter_record(workload)
    if ent is None or ent.get("hbm_bytes_per_launch") is None:
        return None, None, stale
    return ent["hbm_bytes_per_launch"], f"{COUNTER_RECORDS} ({ent.get('kernel', '?')} @ {ent.get('commit', '?')}, sources {ent.get('source_hash')})", False


def valu_roofline(workload, avg_us, per_unit=None):
    """FP64-VALU roofline block from the recorded SQ_INSTS_VALU of the workload's kernel (None without a current record)."""
    ent, stale = counter_record(workload)
    if ent is None or not ent.get("valu_wave_instructions_per_launch"):
        return {"bound": "fp64-valu", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instructions/s", "frac": None,
                "stale": stale, "note": "no PMC record for the current kernel sources (tools/profile.sh)"}
    n = ent["valu_wave_instructions_per_launch"]
    ginstr = n / avg_us / 1e3
    blk = {"bound": "fp64-valu", "achieved": round(ginstr, 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instructions/s", "frac": round(ginstr / VALU_PEAK_GINSTR, 4),
           "valu_wave_instructions_per_launch": n,
           "source": f"{COUNTER_RECORDS} ({ent.get('kernel', '?')} @ {ent.get('commit', '?')}, sources {ent.get('source_hash')})"}
    if per_unit:
        blk[per_unit[0]] = round(n * 64 / per_unit[1], 1)
    return blk


def cpu_leg(a, plan, stack, icrf, diff, rows, n, H, W, with_std, corr):
    """NumPy oracle (a port of the reference's merge arithmetic, oracle/hdr_oracle.py) on rows [0, rows) of the bench stack
    - plus one halo row below when dark maps are on, so that the 3 x 3 medians of the last compared row see their true
    neighbours - single thread, timed; its output is the parity check of the GPU result the timed launches produced."""
    from oracle import hdr_oracle as orc
    hot = corr in (True, "hot")
    flat_on = corr in (True, "flat")
    band = min(rows + (1 if hot else 0), H)
    host = lambda x: x[:band].cpu().numpy()                                              # noqa: E731
    fh = [host(f) for f in stack["frames"]]
    sh = None if stack["stds"] is None else [host(s_) for s_ in stack["stds"]]
    kw = {}
    if hot:
        kw.update(darks=[orc.unit_from_u8(host(d)) for d in stack["darks"]], dark_threshold=DARK_THR, median_k=3)
    if flat_on:
        fk = stack["kw"]
        kw.update(flat=orc.unit_from_u8(host(stack["flat"])), flat_std=host(stack["flat_std"]), ff_mean=np.asarray(fk["ff_mean"]),
                  ff_std_mean=np.asarray(fk["ff_std_mean"]))
    t0 = time.perf_counter()
    ref = orc.merge(fh, stack["t"], icrf, diff if with_std else None, stds=sh, **kw)
    dt = time.perf_counter() - t0
    plan.launch()
    torch.cuda.synchronize()

    def max_rel(got, want):
        got, want = got[:rows], want[:rows]
        err = np.abs(got - want)
        den = np.abs(want)
        nz = den > 0                                     # radiance is exactly 0 where every frame has DN 0 (ICRF[0] = 0)
        return float(max(np.max(err[nz] / den[nz]) if nz.any() else 0.0, np.max(err[~nz]) if (~nz).any() else 0.0))
    rv = max_rel(plan.outputs["val"][:band].cpu().numpy(), ref["val_ff"] if flat_on else ref["val"])
    rs = None
    if with_std:
        rs = max_rel(plan.outputs["std"][:band].cpu().numpy(), ref["std_ff"] if flat_on else ref["std"])
    tol_v = 1e-11 if "f64" in a.workload else 1e-12
    cpu = {"value": round(band * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": f"rows 0..{band - 1} of the bench stack ({n}x{band}x{W}x3, {dt:.1f} s), NumPy oracle, 1 thread of "
                     f"{os.cpu_count()} host cores" + ("; rows 0.." + str(rows - 1) + " compared" if band != rows else ""),
           "gpu_vs_oracle_max_rel_err": rv, "gpu_vs_oracle_max_rel_err_std": rs,
           "parity_ok": bool(rv <= tol_v and (rs is None or rs <= 1e-9))}
    threads = min(16, os.cpu_count() or 1)          # the one-GPU box's CPU share
    if not corr and "f64" not in a.workload:
        # the package's own explicit host backend (Measurand(use_cupy=False): libhdrmerge_host.so, C++ / OpenMP) on the same rows - a product
        # path, reported beside the oracle's number and checked against the same oracle output
        from camera_linearity_amd.measurand import _HOST_ENGINE as heng
        hf = [torch.from_numpy(x) for x in fh]
        hs = None if sh is None else [torch.from_numpy(x) for x in sh]
        t1 = time.perf_counter()
        hout = heng.merge(hf, stack["t"], icrf, diff if with_std else None, hs)
        dt_h = time.perf_counter() - t1
        cpu["host_backend"] = {"value": round(band * W / dt_h / 1e6, 4), "unit": "Mpix/s", "cores": threads, "kind": "product host build (libhdrmerge_host.so, OpenMP)",
                               "max_rel_err_vs_oracle": max_rel(hout["val"].numpy(), ref["val"]), "sample": f"same rows ({dt_h:.2f} s)"}
    if threads > 1 and not corr:
        dt_t = cpu_baseline_threaded(stack["frames"], stack["t"], icrf, diff if with_std else None, stack["stds"], rows, threads)
        cpu["threaded"] = {"value": round(rows * W / dt_t / 1e6, 4), "unit": "Mpix/s", "cores": threads,
                           "sample": f"same rows, 128-row bands on a {threads}-thread pool ({dt_t:.1f} s)"}
    return cpu


def row_tile_workload(a, dev, rank, world, dist):
    """BASELINE.json configs[3] as written: ONE 15-frame 8192 x 8192 x 3 stack cut into 8 row tiles of 1024 rows; the 8 tiles
    are dealt to the ranks (rank r merges tiles r, r + N, ...; N = 1 merges all 8 back to back), each tile is one fused
    launch on that rank's GPU, no data-path collective. A step = every rank merges all its tiles once; value = whole-image
    Mpix per second (total work is fixed: "strong" scaling). After the timed region the image is assembled ONCE on rank 0
    (one image in POSIX shared memory that every rank page-locks and copies its tiles into with asynchronous D2H copies on a
    side stream; the gloo group carries a name and a barrier, no pixels) - reported as `assembly_ms`, never part of `value`."""
    from camera_linearity_amd import parallel
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
    n, H, W, n_tiles = 15, 8192, 8192, 8
    with_std = a.workload == "cfg4std"
    icrf, diff = synthetic_icrf()
    tiles = parallel.RowTileSet(H, n_tiles, rank, world)
    keep = {}
    for tile in tiles.mine:                             # tile t of the image is generated from seed 7 + t whichever rank owns it
        r0, r1 = tiles.bounds[tile]
        frames, stds, t = synthetic_stack_device(7 + tile, n, r1 - r0, W, device=dev, with_std=with_std)
        tiles.add_tile(tile, frames, t, icrf, diff if with_std else None, stds, variant=a.variant)
        keep[tile] = (frames, stds, t)
    alg = tiles.algorithmic_bytes                       # this rank's bytes per step
    n_launch = len(tiles.mine)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    t_end = time.perf_counter() + a.prewarm_s
    while time.perf_counter() < t_end:
        for _ in range(10):
            tiles.launch()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        tiles.launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        tiles.launch()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / a.steps / max(n_launch, 1)
    elapsed, ranks = rank_report(dist, a, dev, elapsed, avg_us)
    # host-side assembly of the image (once, untimed for `value`)
    # the CPU group of the assembly (segment name, shape agreement, barrier): a gloo group inside the nccl world - also at world
    # size 1 under --force-dist, so that this combination runs on a one-GPU box
    gloo = dist.new_group(backend="gloo") if dist is not None else None
    barrier()
    ta = time.perf_counter()
    val, std = tiles.assemble(group=gloo, dst=0)        # the first call also page-locks the 1.6 GB (3.2 GB with std) image buffer
    assembly_first_ms = (time.perf_counter() - ta) * 1e3
    barrier()
    ta = time.perf_counter()
    val, std = tiles.assemble(group=gloo, dst=0)        # steady state: D2H of every tile into its rows of the pinned image (+ the sends)
    assembly_ms = (time.perf_counter() - ta) * 1e3
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import hdr_oracle as orc
        rows = a.cpu_rows if a.cpu_rows > 0 else (192 if with_std else 384)
        tile = n_tiles // 2 if n_tiles // 2 in tiles.mine else tiles.mine[0]           # an interior tile (row0 = 4096)
        frames, stds, t = keep[tile]
        fh = [f[:rows].cpu().numpy() for f in frames]
        sh = None if stds is None else [s_[:rows].cpu().numpy() for s_ in stds]
        c0 = time.perf_counter()
        ref = orc.merge(fh, t, icrf, diff if with_std else None, stds=sh)
        dt = time.perf_counter() - c0
        r0 = tiles.bounds[tile][0]
        got = val[r0:r0 + rows].numpy()                                                 # out of the ASSEMBLED image
        den = np.abs(ref["val"])
        nz = den > 0
        rv = float(np.max(np.abs(got - ref["val"])[nz] / den[nz]))
        rs = None
        if with_std:
            rs = float(np.max(np.abs(std[r0:r0 + rows].numpy() - ref["std"]) / np.abs(ref["std"])))
        cpu = {"value": round(rows * W / dt / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
               "sample": f"rows {r0}..{r0 + rows - 1} of the image = the first {rows} rows of tile {tile} ({n}x{rows}x{W}x3, {dt:.1f} s), "
                         f"NumPy oracle, 1 thread of {os.cpu_count()} host cores; compared with the same rows of the assembled image",
               "gpu_vs_oracle_max_rel_err": rv, "gpu_vs_oracle_max_rel_err_std": rs, "parity_ok": bool(rv <= 1e-12 and (rs is None or rs <= 1e-9))}
    if rank == 0:
        plan0 = tiles.plans[tiles.mine[0]]
        achieved = alg / n_launch / avg_us / 1e3
        traffic, traffic_src, traffic_stale = measured_traffic("cfg4tilestd" if with_std else "cfg4tile")   # one launch = one tile
        line = {"metric": "HDR-merged Mpix/s (node)", "value": round(a.steps * H * W / elapsed / 1e6, 1), "unit": "Mpix/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 5), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"ONE {n}x{H}x{W}x3 uint8 exposure stack in {n_tiles} row tiles of {H // n_tiles} rows dealt to {world} GPU(s), "
                                       + ("float64 std propagation" if with_std else "val-only merge") + ", host-side assembly (untimed)",
                           "name": a.workload, "frames": n, "height": H, "width": W, "channels": 3, "row_tiles": n_tiles,
                           "parallelism": f"row tiles: {n_launch} per GPU, no collective on the data path", "variant": a.variant},
                "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                             "kernel": plan0.kernels,
                             "algorithmic_bytes_per_launch": alg // n_launch, "avg_launch_us": round(avg_us, 2)},
                "ranks": ranks,
                "assembly_ms": round(assembly_ms, 1), "assembly_first_ms": round(assembly_first_ms, 1),
                "assembly": f"{n_tiles} tiles -> async D2H on a side stream into their rows of one {H}x{W}x3 float64 image"
                            + (" (+ std)" if with_std else "") + f" on rank 0 [{tiles.assembly_path}]",
                "cpu_baseline": cpu}
        print(json.dumps(line))
    del val, std
    tiles.close()
    if dist is not None:
        dist.destroy_process_group()


def producer_workload(a, dev, rank=0, world=1, dist=None):
    """The SURVEY.md 8(f) rows as bench workloads: same timing discipline as the merge (pre-warm, warm-up, K launches
    between two events) and a cpu_baseline leg = the oracle on a bounded sample (N = 1 only). At N > 1 every rank runs its own
    replica of the workload (no exchange step: "replicas only"). Rank 0 prints one JSON line."""
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
        kernel, bound_note = "k_welford", "FP64 VALU from ~12 frames per launch up (DESIGN.md 4.4): roofline_valu is the bound that binds at 32 frames per launch"
        valu_unit = ("instructions_per_element_frame", H * W * 3 * K)
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
        kernel, bound_note = "k_energy_pixel", "FP64 VALU (DESIGN.md 4.4): roofline_valu is the relevant bound, the HBM figure is not"
        valu_unit = ("instructions_per_pixel_pair_candidate", X * Y * (N * (N - 1) // 2) * B)
    t_end = time.perf_counter() + a.prewarm_s
    while time.perf_counter() < t_end:
        launch()
        torch.cuda.synchronize()
    for _ in range(warmup):
        launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        launch()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / steps
    elapsed, ranks = rank_report(dist, a, dev, elapsed, avg_us)
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return
    cpu = None
    if not a.no_cpu_baseline and world == 1:
        from oracle import hdr_oracle as orc
        if a.workload == "welford":
            rows_c = 2048
            sample = [f[:rows_c].cpu().numpy() for f in clip]
            c0 = time.perf_counter()
            want = orc.welford_state(sample, None, True)
            dt = time.perf_counter() - c0
            # parity: ONE clean fold of the same band (fresh zero state, count 0) through the timed kernel against the oracle's state
            gm = torch.zeros((rows_c, W, 3), dtype=torch.float64, device=dev)
            g2 = torch.zeros_like(gm)
            engine.welford_update([f[:rows_c] for f in clip], 0, gm, g2)
            torch.cuda.synchronize()

            def rel(got, ref):
                den = np.maximum(np.abs(ref), 1e-300)
                return float(np.max(np.abs(got - ref) / den))
            e_mean, e_m2 = rel(gm.cpu().numpy(), want[0]), rel(g2.cpu().numpy(), want[1])
            cpu = {"value": round(rows_c * W * K / dt / 1e6, 3), "unit": unit, "cores": 1, "kind": "port",
                   "sample": f"{K} frames of {rows_c}x{W}x3 ({dt:.1f} s), NumPy oracle, 1 thread; the same band folded once on the GPU is compared with it",
                   "gpu_vs_oracle_max_rel_err": e_mean, "gpu_vs_oracle_max_rel_err_m2": e_m2, "parity_ok": bool(e_mean <= 1e-13 and e_m2 <= 1e-12)}
            # the package's own host build of the same entry point (libhdrmerge_host.so, OpenMP) on the same band: bit-identical to the device
            from camera_linearity_amd.measurand import _HOST_ENGINE as heng
            hm_, h2_ = torch.zeros((rows_c, W, 3), dtype=torch.float64), torch.zeros((rows_c, W, 3), dtype=torch.float64)
            hs = [torch.from_numpy(x) for x in sample]
            c0 = time.perf_counter()
            heng.welford_update(hs, 0, hm_, h2_)
            dt_h = time.perf_counter() - c0
            cpu["host_backend"] = {"value": round(rows_c * W * K / dt_h / 1e6, 3), "unit": unit, "cores": min(16, os.cpu_count() or 1),
                                   "kind": "product host build (libhdrmerge_host.so, OpenMP)",
                                   "bit_identical_to_gpu": bool(torch.equal(hm_, gm.cpu()) and torch.equal(h2_, g2.cpu()))}
        else:
            reps = 10
            c0 = time.perf_counter()
            want = [orc.energy_function(icrfs[b], dn_h, None, 5, 250, t) for b in range(reps)]
            dt = time.perf_counter() - c0
            got = engine.linearity_energy(dn, None, t, icrfs_d, 5, 250).cpu().numpy()[:reps]
            worst = float(np.max(np.abs(got - np.asarray(want)) / np.abs(np.asarray(want))))
            cpu = {"value": round(reps / dt, 4), "unit": unit, "cores": 1, "kind": "port",
                   "sample": f"{reps} of the 75 candidates on the full 1024x1024x7 stack ({dt:.1f} s), NumPy oracle, 1 thread; the GPU energies of the same candidates are compared with it",
                   "gpu_vs_oracle_max_rel_err": worst, "parity_ok": bool(worst <= 1e-10)}
            from camera_linearity_amd.measurand import _HOST_ENGINE as heng
            c0 = time.perf_counter()
            got_h = heng.linearity_energy(torch.from_numpy(dn_h), None, t, icrfs[:reps], 5, 250).numpy()
            dt_h = time.perf_counter() - c0
            cpu["host_backend"] = {"value": round(reps / dt_h, 3), "unit": unit, "cores": min(16, os.cpu_count() or 1),
                                   "kind": "product host build (libhdrmerge_host.so, OpenMP)",
                                   "max_rel_diff_to_gpu": float(np.max(np.abs(got_h - got) / np.abs(got)))}
    scale_u = 1e6 if unit.startswith("M") else 1.0
    traffic, traffic_src, traffic_stale = measured_traffic(a.workload)
    line = {"metric": metric, "value": round(world * steps * units / elapsed / scale_u, 2), "unit": unit, "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": cfg,
            "roofline": {"bound": "hbm", "achieved": round(alg / avg_us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg / avg_us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "kernel": kernel, "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(avg_us, 2), "note": bound_note},
            "roofline_valu": valu_roofline(a.workload, avg_us, valu_unit), "ranks": ranks, "cpu_baseline": cpu}
    print(json.dumps(line), flush=True)


VALU_PEAK_GINSTR = 614.4        # FP64 vector issue rate of the part in 64-lane wave-instructions: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles
                                # (= 78.6 TFLOP/s of FP64 FMA, half of the guide's 157.3 TFLOP/s FP32 vector peak)


def linearity_workload(a, dev, rank=0, world=1, dist=None):
    """SURVEY.md 8(f)-1 as a bench workload: ExposureSeries.process_linearity (modules/exposure_series.py:421-446) on a device-resident
    linearized series - 7 float64 frames of 4096 x 4096 x 3 (`linearitystd`: + 7 float64 std frames), the 15 exposure pairs with ratio
    >= 0.1. A step = ONE hm_pairs_statistics launch: apply_thresholds on every frame in place (fused into the loads), absolute and
    relative difference of every pair, (weighted) NaN-ignoring mean / std / error per channel - what process_linearity runs per call.
    Two rooflines: HBM (every frame byte read once) and FP64 VALU (the kernel's binding resource). CPU leg = the NumPy oracle
    (apply_thresholds + compute_difference + dimension_statistics per pair) on a row band; the GPU statistics of the same band are
    checked against it. At N > 1 every rank runs its own series (replicas only: there is no exchange step)."""
    from camera_linearity_amd import engine
    from camera_linearity_amd.exposure_series import map_linearity_limits
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
    n, H, W = 7, 4096, 4096
    use_std = a.workload == "linearitystd"
    icrf, diff = synthetic_icrf()
    frames, stds, t = synthetic_stack_device(7 + rank, n, H, W, device=dev, with_std=use_std)
    lo, hi = map_linearity_limits(5, 5, icrf)
    pairs = [(i, j, float(t[i] / t[j])) for i in range(n) for j in range(n) if i < j and t[i] / t[j] >= 0.1]   # exposure_series.py:283-304
    rows_cpu = a.cpu_rows if a.cpu_rows > 0 else 192

    def linearized(sl):
        vals, sds = [], []
        for f, sd in zip(frames, stds if use_std else [None] * n):
            v, s_ = engine.linearize(f[sl], None if sd is None else sd[sl], icrf, diff if use_std else None)
            vals.append(v)
            sds.append(s_)
        return vals, (sds if use_std else None)
    # the band the CPU leg compares, linearized BEFORE anything is thresholded in place
    cpu = None
    band_stats = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        bv, bs = linearized(slice(0, rows_cpu))
        band_host = ([v.cpu().numpy() for v in bv], None if bs is None else [x.cpu().numpy() for x in bs])
        band_stats = engine.pairs_statistics(bv, bs, pairs, to_host=True, thresholds=(lo, hi))
        del bv, bs
    vals, sds = linearized(slice(None))
    del frames, stds
    launch = lambda: engine.pairs_statistics(vals, sds, pairs, thresholds=(lo, hi))      # noqa: E731
    launch()                                            # the first call thresholds the frames (writes NaNs back); later calls find them thresholded
    torch.cuda.synchronize()
    E = H * W * 3
    alg = n * 8 * (2 if use_std else 1) * E             # every frame (and std) byte once
    steps, warmup = min(a.steps, 100), min(a.warmup, 10)
    # untimed clock pre-warm: this kernel is FP64-VALU bound, so its time IS the shader clock - and after the seconds of host-side set-up
    # above the clock needs ~2 s of load to settle (0.5 s of pre-warm measured 2 790-3 575 us per launch, 3 s 2 457-2 460 us on the same
    # box in the same call: tools/lin_var.sh, profiles/r04_lin_prewarm.log). The HBM-bound merges settle within the default 0.5 s.
    t_end = time.perf_counter() + max(a.prewarm_s, 3.0)
    while time.perf_counter() < t_end:
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
    for _ in range(warmup):
        launch()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        launch()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / steps
    elapsed, ranks = rank_report(dist, a, dev, elapsed, avg_us)
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return
    if band_stats is not None:
        from oracle import hdr_oracle as orc
        hv, hs = band_host
        c0 = time.perf_counter()
        th = [orc.apply_thresholds(v, None if hs is None else hs[i], lo, hi) for i, v in enumerate(hv)]
        worst = 0.0
        with np.errstate(all="ignore"):
            for k, (i, j, m) in enumerate(pairs):
                d = orc.compute_difference(th[i][0], th[i][1], th[j][0], th[j][1], m)
                for kind, (dv, ds) in enumerate(((d[0], d[1]), (d[2], d[3]))):
                    st = orc.dimension_statistics(dv, ds, axis=(0, 1))
                    got = band_stats[k][kind]
                    for key in ("mean", "std") + (("error",) if use_std else ()):
                        want = np.asarray(st[key], dtype=np.float64)
                        g_ = np.asarray(got[key], dtype=np.float64)
                        ok = np.isfinite(want)
                        if ok.any():
                            worst = max(worst, float(np.max(np.abs(g_[ok] - want[ok]) / np.maximum(np.abs(want[ok]), 1e-300))))
        dt = time.perf_counter() - c0
        cpu = {"value": round(len(pairs) * rows_cpu * W / dt / 1e6, 4), "unit": "Mpix-pairs/s", "cores": 1, "kind": "port",
               "sample": f"rows 0..{rows_cpu - 1} of the series ({n}x{rows_cpu}x{W}x3 float64" + (" + std" if use_std else "") + f", {len(pairs)} pairs, {dt:.1f} s): "
                         "apply_thresholds + compute_difference + dimension_statistics per pair, NumPy oracle, 1 thread of " + f"{os.cpu_count()} host cores",
               "gpu_vs_oracle_max_rel_err": worst, "parity_ok": bool(worst <= 1e-10)}
    valu_block = valu_roofline(a.workload, avg_us, ("instructions_per_element_pair", len(pairs) * E))
    traffic, traffic_src, traffic_stale = measured_traffic(a.workload)
    line = {"metric": "linearity-compared Mpix-pairs/s (process_linearity: 15 exposure pairs of 7 frames of 4096x4096x3)", "value": round(world * steps * len(pairs) * H * W / elapsed / 1e6, 1),
            "unit": "Mpix-pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n} linearized float64 frames of {H}x{W}x3" + (" + float64 std" if use_std else "") + f" per GPU, {len(pairs)} exposure pairs, thresholds fused, "
                                   + ("weighted " if use_std else "") + "statistics of the absolute and relative differences per channel", "name": a.workload, "frames": n, "pairs": len(pairs),
                       "height": H, "width": W, "channels": 3, "parallelism": "replicas only (no exchange step)"},
            "roofline": {"bound": "hbm", "achieved": round(alg / avg_us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg / avg_us / 1e3 / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "kernel": "k_pairs_stats_lds",
                         "algorithmic_bytes_per_launch": alg, "avg_launch_us": round(avg_us, 2),
                         "note": "read-once bytes; the kernel is FP64-VALU bound (roofline_valu)"},
            "roofline_valu": valu_block, "ranks": ranks, "cpu_baseline": cpu}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + ["cfg4", "cfg4std", "welford", "energy", "linearity", "linearitystd"])
    ap.add_argument("--stacks", type=int, default=0, help="distinct resident stacks merged round-robin (0 = 4 for cfg2 / cfg2rand, 2 for the cfg3 family, 1 otherwise)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal only: initialise the process group even at world size 1")
    ap.add_argument("--hot-density", type=float, default=1e-4, help="fraction of hot elements in the synthetic dark maps (cfg3 / cfg3hot)")
    ap.add_argument("--shared-dark", action="store_true", help="cfg3 / cfg3hot: ONE dark map shared by the 7 frames (round 2's setup: the map is read once and a hot "
                                                               "element is hot in every frame) instead of one map per frame (SURVEY 8d: d * N bytes)")
    ap.add_argument("--no-hot-queue", action="store_true", help="A/B: run the hot-pixel pass without its queue workspace (round 2's one-element-per-wave pass)")
    ap.add_argument("--prewarm-s", type=float, default=0.5, help="seconds of untimed launches before the warm-up steps")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the stack the CPU baseline merges (0 = per workload: 4096 val-only, 512 with std / corrections)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    late_imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.share_device:
        local_rank = 0
    have = torch.cuda.device_count()
    if have <= local_rank or (not a.share_device and have < world):
        # one clear line per rank (every rank of the run exits, none waits in a rendezvous) instead of a set_device traceback
        print(f"bench.py: rank {rank}: --gpus {a.gpus} needs GPU {local_rank} of {world} but this node shows {have} GPU(s) "
              "(--share-device puts every rank on GPU 0: rehearsal only)", file=sys.stderr, flush=True)
        sys.exit(3)
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
        producer_workload(a, dev, rank, world, dist)
        return
    if a.workload in ("linearity", "linearitystd"):
        linearity_workload(a, dev, rank, world, dist)
        return
    from camera_linearity_amd import engine
    from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf, synthetic_flat_dark

    if a.workload in ("cfg4", "cfg4std"):             # configs[3] as specified: ONE 15 x 8192 x 8192 x 3 image in 8 row tiles
        row_tile_workload(a, dev, rank, world, dist)
        return
    n, H, W, with_std, corr = WORKLOADS[a.workload]
    icrf, diff = synthetic_icrf()
    # config 2: four stacks so that no launch finds the previous one's inputs in the 256 MB Infinity Cache; config 3's stacks (4 GB of inputs
    # each) cannot be cached anyway - two of them average the placement lottery (the same kernel on the same data takes 746-805 us depending on
    # where its buffers landed) instead of reporting one draw
    rotate = a.stacks if a.stacks > 0 else (4 if a.workload in ("cfg2", "cfg2rand", "cfg2smooth") else 2 if a.workload in ("cfg3", "cfg3std", "cfg3flat", "cfg3hot") else 1)
    if a.workload == "cfg5":
        rotate = 1

    def build_stack(seed):
        frames, stds, t = synthetic_stack_device(seed, n, H, W, device=dev, with_std=with_std, smooth=a.workload == "cfg2smooth")
        if "f64" in a.workload:
            frames = [engine.u8_to_unit(f) for f in frames]
        if a.workload == "cfg2rand":
            gen = torch.Generator(device=dev).manual_seed(seed)
            frames = [torch.randint(0, 256, f.shape, dtype=torch.uint8, device=dev, generator=gen) for f in frames]
        kw = {}
        extra = {}
        if corr:
            flat, flat_std, dark = synthetic_flat_dark(seed, H, W, device=dev, hot_density=a.hot_density)
            darks = [dark] * n
            if not a.shared_dark:            # one dark map per frame (the reference picks a dark frame per exposure, image_set.py:157-198)
                darks = [dark] + [synthetic_flat_dark(seed + 17 * i, H, W, device=dev, hot_density=a.hot_density)[2] for i in range(1, n)]
            extra.update(flat=flat, flat_std=flat_std, darks=darks)
            x0, x1, y0, y1 = engine.flat_roi_bounds(H, W, 0.2)
            if corr in (True, "flat"):
                kw.update(flat=flat, flat_std=flat_std, ff_mean=engine.roi_mean(flat, x0, x1, y0, y1).cpu().numpy(),
                          ff_std_mean=engine.roi_mean(flat_std, x0, x1, y0, y1).cpu().numpy())
            if corr in (True, "hot"):
                kw.update(darks=darks, dark_min=[engine.dark_min_dn(1.0, DARK_THR)] * n, median_k=3, hot_queue=not a.no_hot_queue)
        plan = engine.plan_merge(frames, t, icrf, diff if with_std else None, stds, variant=a.variant, **kw)
        return plan, dict(frames=frames, stds=stds, t=t, kw=kw, **extra)

    # R distinct resident stacks, merged round-robin: consecutive launches never re-read the previous launch's inputs, so no
    # input byte can come out of the 256 MB Infinity Cache (one config-2 stack is 352 MB in + 403 MB out)
    plans, stacks = [], []
    for k in range(rotate):
        p_, s_ = build_stack(7 + rank + 100 * k)
        plans.append(p_)
        stacks.append(s_)
    plan, stack0 = plans[0], stacks[0]
    alg_bytes = plan.algorithmic_bytes
    launches_per_step = 1
    if a.workload == "cfg5":                 # 8 distinct resident stacks per GPU; a step = 8 launches (one per stack)
        launches_per_step = 8
        for k in range(1, launches_per_step):
            plans.append(build_stack(100 * k + 7 + rank)[0])

    counter = [0]

    def step():
        if launches_per_step > 1:
            for p_ in plans:
                p_.launch()
        else:
            plans[counter[0] % len(plans)].launch()
            counter[0] += 1

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
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    # HIP events on the launch stream (torch's current stream is the one hm_merge is given), bracketing the
    # K back-to-back launches of the timed region: average launch duration = event span / K. (An event pair
    # around EVERY launch would put a barrier packet between consecutive kernels and measure a different,
    # serialised pipeline.) A second pass after the timed region records per-launch events for min/max only.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_us = ev0.elapsed_time(ev1) * 1e3 / a.steps / launches_per_step      # per hm_merge call
    # secondary figure: the same number of launches re-merging ONE stack (what round 1 reported as the headline)
    same_us = None
    if len(plans) > 1 and launches_per_step == 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            plan.launch()
        e0.record()
        for _ in range(a.steps):
            plan.launch()
        e1.record()
        torch.cuda.synchronize()
        same_us = e0.elapsed_time(e1) * 1e3 / a.steps
    per = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(a.steps, 20))]
    for e0, e1 in per:
        e0.record()
        plan.launch()
        e1.record()
    torch.cuda.synchronize()
    kernel_us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in per]
    # the box's copy rate (the library's nontemporal 16-byte copy kernel between two output buffers, after the timed region): the merge's
    # traffic is ~half reads, ~half writes, and a plain copy is what the memory system sustains for such a mix
    copy_gbps = None
    pair_ = None
    if len(plans) > 1 and "val" in plans[0].outputs and "val" in plans[1].outputs:
        pair_ = (plans[0].outputs["val"], plans[1].outputs["val"])
    elif "val" in plans[0].outputs and "std" in plans[0].outputs:
        pair_ = (plans[0].outputs["val"], plans[0].outputs["std"])
    if rank == 0 and pair_ is not None:
        from camera_linearity_amd import _native as nat
        src, dst = pair_
        nbytes = src.numel() * 8
        st_ = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(20):
            nat.check(nat.lib.hm_debug_copy_probe(src.data_ptr(), dst.data_ptr(), nbytes, st_), "copy probe")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(100):
            a_, b_ = (src, dst) if k % 2 == 0 else (dst, src)
            nat.check(nat.lib.hm_debug_copy_probe(a_.data_ptr(), b_.data_ptr(), nbytes, st_), "copy probe")
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 2 * nbytes / (e0.elapsed_time(e1) * 1e-3 / 100) / 1e9
        for p_ in plans[:2]:                                       # restore the outputs the CPU leg will check
            p_.launch()
        torch.cuda.synchronize()
    elapsed, ranks = rank_report(dist, a, dev, elapsed, avg_us)

    # CPU leg + parity check of the timed configuration against the oracle on a row band (not timed on the GPU side)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        frames, stds, t = stack0["frames"], stack0["stds"], stack0["t"]
        rows = min(a.cpu_rows if a.cpu_rows > 0 else (4096 if not with_std and not corr else 512), H)
        if "f64" in a.workload:
            rows = min(rows, 1024)
        cpu = cpu_leg(a, plan, stack0, icrf, diff, rows, n, H, W, with_std, corr)

    if rank == 0:
        achieved = alg_bytes / avg_us / 1e3          # GB/s
        traffic, traffic_src, traffic_stale = measured_traffic(a.workload, plan.kernels)
        mpix = world * a.steps * launches_per_step * H * W / elapsed / 1e6
        line = {
            "metric": "HDR-merged Mpix/s (node)", "value": round(mpix, 1), "unit": "Mpix/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n}x{H}x{W}x3 uint8 exposure stack per GPU, 256-entry ICRF LUT, "
                                   + ("float64 std + dark hot-pixel maps + flat field" if corr is True else
                                      ("float64 std + " + ("flat field" if corr == "flat" else "dark hot-pixel maps") if corr else
                                       ("float64 std propagation" if with_std else "val-only merge")))
                                   + " -> float64 radiance" + (" + uncertainty" if with_std else "")
                                   + (f"; {len(plans)} distinct resident stacks merged round-robin (cold inputs)" if len(plans) > 1 and launches_per_step == 1 else ""),
                       "name": a.workload, "frames": n, "height": H, "width": W, "channels": 3, "resident_stacks": len(plans),
                       "hot_density": a.hot_density if corr in (True, "hot") else None,
                       "dark_maps": (("1 shared by the frames" if a.shared_dark else f"{n} distinct") + ("" if not a.no_hot_queue else ", no queue workspace"))
                       if corr in (True, "hot") else None,
                       "parallelism": f"independent stacks x{world * max(launches_per_step, 1)} per step, no collective", "variant": a.variant},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "kernel": plan.kernels, "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": round(avg_us, 2),
                         "same_stack_avg_launch_us": None if same_us is None else round(same_us, 2),
                         "same_stack_frac": None if same_us is None else round(alg_bytes / same_us / 1e3 / HBM_PEAK_GBPS, 4),
                         "isolated_launch_us_min_median": [round(float(np.min(kernel_us)), 2), round(float(np.median(kernel_us)), 2)],
                         "copy_GBps": None if copy_gbps is None else round(copy_gbps, 1),
                         "frac_of_copy": None if copy_gbps is None else round(achieved / copy_gbps, 4)},
            "ranks": ranks,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
