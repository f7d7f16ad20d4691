#!/usr/bin/env python3
"""ONE large exposure stack merged by N GPUs of a node: row tiles dealt to the ranks, no collective on the data path.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        examples/merge_one_image_on_n_gpus.py [--frames 15 --height 8192 --width 8192 --tiles 8 --std]

Every output element of the merge depends on the N input elements at the same position only (modules/exposure_series.py:388-389), so the
image is cut into contiguous row tiles (a tile is one contiguous byte range of every frame), rank r merges tiles r, r + world, ... on its own
GPU, and the result is assembled on rank 0 in a POSIX shared-memory image that every rank copies its rows into (parallel.RowTileSet /
SharedHostImage). The process group is gloo: it carries a name, a flag and a barrier - there is no GPU<->GPU traffic, RCCL is not needed.
Hot-pixel medians read a halo row above and below a tile from the neighbouring tile's INPUT rows (`input_rows`), never from another GPU.

The stack here is synthetic (the bench's recipe, every rank generating its own rows) - a real caller reads each tile's rows from its TIFFs
instead (`tiff_io.imread` + slicing).
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from camera_linearity_amd import parallel  # noqa: E402
from camera_linearity_amd.synthetic import synthetic_exposures, synthetic_icrf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=15)
    ap.add_argument("--height", type=int, default=8192)
    ap.add_argument("--width", type=int, default=8192)
    ap.add_argument("--tiles", type=int, default=8)
    ap.add_argument("--std", action="store_true", help="propagate uncertainty (float64 std frames: 8 bytes per element and frame)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: all ranks on GPU 0")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        dist.init_process_group("gloo")                       # CPU group: names, flags, barriers - the data never leaves its GPU
    have = torch.cuda.device_count()
    if have == 0 or (not a.share_device and local >= have):
        sys.exit(f"rank {rank}: needs GPU {local}, the node shows {have}")
    dev = torch.device("cuda", 0 if a.share_device else local)
    torch.cuda.set_device(dev)

    N, H, W, C = a.frames, a.height, a.width, 3
    icrf, diff = synthetic_icrf()
    t = synthetic_exposures(N)
    tiles = parallel.RowTileSet(H, a.tiles, rank=rank, world_size=world, median_k=0, row_elems=W * C)   # (row_elems: odd W * C -> tiles start on even rows)
    gen = np.random.default_rng(7)                            # the same stream on every rank: each cuts its own rows out of the same image
    k = 255.0 / (4.0 * t[N // 2])
    for tile in range(a.tiles):
        r0, r1 = tiles.input_rows(tile)
        rad = gen.random((r1 - r0, W, C)) * 4.0               # (every rank draws every tile's numbers to stay in step; only its own are uploaded)
        if tile not in tiles.mine:
            continue
        frames = [torch.from_numpy(np.clip(np.around(rad * ti * k), 0, 255).astype(np.uint8)).to(dev, non_blocking=True) for ti in t]
        stds = [torch.full((r1 - r0, W, C), 0.004, dtype=torch.float64, device=dev) for _ in t] if a.std else None
        tiles.add_tile(tile, frames, t, icrf, diff if a.std else None, stds)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    group = dist.group.WORLD if world > 1 else None
    tiles.launch()                                            # first pass, untimed: kernels are loaded, the host image is created and page-locked
    tiles.assemble(group=group, dst=0)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    tiles.launch()
    torch.cuda.synchronize()
    t_merge = time.perf_counter() - t0
    val, std = tiles.assemble(group=group, dst=0)
    t_all = time.perf_counter() - t0
    if rank == 0:
        px = H * W
        print(f"{N} x {H} x {W} x {C} in {a.tiles} row tiles on {world} rank(s), second pass: rank 0's tiles merged in {t_merge * 1e3:.2f} ms "
              f"({px / t_merge / 1e6:.0f} Mpix/s if every rank takes as long), image assembled on the host after {t_all * 1e3:.1f} ms via {tiles.assembly_path}; "
              f"val {tuple(val.shape)} mean {float(val.mean()):.6f}" + ("" if std is None else f", std mean {float(std.mean()):.3e}"))
    tiles.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
