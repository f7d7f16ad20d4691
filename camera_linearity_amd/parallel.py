"""Data-parallel sharding of the merge across the GPUs of one node.

The merge is per-pixel independent (modules/exposure_series.py:388-389), so there is NO data-path
collective: one process per GPU (torch.distributed, launched with torch.distributed.run), each rank
merges its own units and results are assembled host-side. Two shardings:

  * row tiles   - one large stack split into contiguous row ranges (HWC row-major => one contiguous
                  byte range per frame). The hot-pixel median needs a floor(k/2)-row halo of INPUT rows
                  from the neighbouring tile; 'reflect' applies only at the true image edges. The two
                  flat-field ROI means are global scalars, computed once and passed by value.
  * whole stacks - a batch of independent stacks dealt round-robin to the ranks.

The only inter-process traffic is the optional result gather over the CPU (gloo) group.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def row_tile_bounds(height: int, world_size: int) -> List[Tuple[int, int]]:
    """Rank r owns rows [r*H/G, (r+1)*H/G) (integer floor): contiguous, disjoint, covering, balanced to 1 row."""
    if height < 0 or world_size < 1:
        raise ValueError("height >= 0 and world_size >= 1 required")
    return [((r * height) // world_size, ((r + 1) * height) // world_size) for r in range(world_size)]


def halo_bounds(row0: int, row1: int, height: int, median_k: int = 0) -> Tuple[int, int]:
    """Input rows a tile must hold: its own rows plus floor(k/2) rows either side, clipped to the image."""
    r = median_k // 2 if median_k else 0
    return max(0, row0 - r), min(height, row1 + r)


def stacks_for_rank(n_stacks: int, rank: int, world_size: int) -> List[int]:
    """Round-robin deal of independent stacks (config 5)."""
    return list(range(rank, n_stacks, world_size))


def merge_row_tile(frames_host: Sequence[np.ndarray], exposures, icrf, icrf_diff=None, stds_host=None,
                   darks_host=None, dark_min=None, median_k: int = 3, flat_host=None, flat_std_host=None,
                   ff_mean=None, ff_std_mean=None, rank: int = 0, world_size: int = 1, device=None):
    """Merge this rank's row tile of a stack that lives in host memory. Uploads only the rows the tile
    needs (tile + halo), launches the fused kernel, returns (row0, row1, val, std) with host arrays."""
    from . import engine
    H = frames_host[0].shape[0]
    r0, r1 = row_tile_bounds(H, world_size)[rank]
    use_hot = darks_host is not None and any(d is not None for d in darks_host)
    b0, b1 = halo_bounds(r0, r1, H, median_k if use_hot else 0)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    up = lambda a, lo, hi: torch.as_tensor(np.ascontiguousarray(a[lo:hi]), device=device)   # noqa: E731
    frames = [up(f, b0, b1) for f in frames_host]
    stds = None if stds_host is None else [up(s, b0, b1) for s in stds_host]
    darks = None if not use_hot else [None if d is None else up(d, b0, b1) for d in darks_host]
    kw = {}
    if flat_host is not None:
        kw.update(flat=up(flat_host, r0, r1), ff_mean=ff_mean)
        if stds is not None:
            kw.update(flat_std=up(flat_std_host, r0, r1), ff_std_mean=ff_std_mean)
    if r1 == r0:
        empty = np.empty((0,) + tuple(frames_host[0].shape[1:]))
        return r0, r1, empty, (None if stds is None else empty.copy())
    out = engine.merge(frames, exposures, icrf, icrf_diff, stds, darks=darks, dark_min=dark_min, median_k=median_k,
                       height=H, row0=r0, rows=r1 - r0, buf_row0=b0, **kw)
    val = out["val"].cpu().numpy()
    std = out["std"].cpu().numpy() if "std" in out else None
    return r0, r1, val, std


def gather_row_tiles(local_val: np.ndarray, local_std: Optional[np.ndarray], group=None, dst: int = 0):
    """Host-side assembly of the tiles on rank `dst` through the CPU process group (gloo): tiles are
    concatenated in rank order. Returns (val, std) on dst, (None, None) elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    payload = (local_val, local_std)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst, group=group)
    if rank != dst:
        return None, None
    val = np.concatenate([g[0] for g in gathered], axis=0)
    std = None if gathered[0][1] is None else np.concatenate([g[1] for g in gathered], axis=0)
    return val, std
