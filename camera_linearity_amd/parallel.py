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


def tiles_for_rank(n_tiles: int, rank: int, world_size: int) -> List[int]:
    """Row tiles of ONE image dealt to the ranks (config 4: 8 tiles; rank r takes tiles r, r + G, ...)."""
    return list(range(rank, n_tiles, world_size))


def _pinned_like(a: np.ndarray) -> torch.Tensor:
    return torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, pin_memory=torch.cuda.is_available())


def merge_row_tile(frames_host: Sequence[np.ndarray], exposures, icrf, icrf_diff=None, stds_host=None,
                   darks_host=None, dark_min=None, median_k: int = 3, flat_host=None, flat_std_host=None,
                   ff_mean=None, ff_std_mean=None, rank: int = 0, world_size: int = 1, device=None,
                   tile: Optional[Tuple[int, int]] = None):
    """Merge this rank's row tile of a stack that lives in host memory. Only the rows the tile needs (tile + halo) travel:
    they are staged in pinned buffers and copied with asynchronous H2D copies on the current stream, the fused kernel is
    launched behind them, and the result comes back through pinned buffers with asynchronous D2H copies (one
    synchronisation at the end). Returns (row0, row1, val, std) with host arrays. `tile` overrides the (row0, row1) that
    row_tile_bounds() gives this rank (a rank that owns several tiles calls once per tile)."""
    from . import engine
    H = frames_host[0].shape[0]
    r0, r1 = row_tile_bounds(H, world_size)[rank] if tile is None else tile
    use_hot = darks_host is not None and any(d is not None for d in darks_host)
    b0, b1 = halo_bounds(r0, r1, H, median_k if use_hot else 0)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if r1 == r0:
        empty = np.empty((0,) + tuple(frames_host[0].shape[1:]))
        return r0, r1, empty, (None if stds_host is None else empty.copy())
    staged = []

    def up(a, lo, hi):
        h = _pinned_like(a[lo:hi])
        h.numpy()[...] = a[lo:hi]                        # pageable -> pinned (the only host copy), then DMA
        staged.append(h)
        return h.to(device, non_blocking=True)
    frames = [up(f, b0, b1) for f in frames_host]
    stds = None if stds_host is None else [up(s, b0, b1) for s in stds_host]
    darks = None if not use_hot else [None if d is None else up(d, b0, b1) for d in darks_host]
    kw = {}
    if flat_host is not None:
        kw.update(flat=up(flat_host, r0, r1), ff_mean=ff_mean)
        if stds is not None:
            kw.update(flat_std=up(flat_std_host, r0, r1), ff_std_mean=ff_std_mean)
    out = engine.merge(frames, exposures, icrf, icrf_diff, stds, darks=darks, dark_min=dark_min, median_k=median_k,
                       height=H, row0=r0, rows=r1 - r0, buf_row0=b0, **kw)
    h_val = torch.empty(out["val"].shape, dtype=torch.float64, pin_memory=True)
    h_val.copy_(out["val"], non_blocking=True)
    h_std = None
    if "std" in out:
        h_std = torch.empty(out["std"].shape, dtype=torch.float64, pin_memory=True)
        h_std.copy_(out["std"], non_blocking=True)
    torch.cuda.current_stream(device).synchronize()
    return r0, r1, h_val.numpy(), (None if h_std is None else h_std.numpy())


class RowTileSet:
    """The row tiles of ONE large image that this rank owns, resident on its GPU (config 4: a 15 x 8192 x 8192 x 3
    stack cut into 8 tiles of 1024 rows). `launch()` enqueues one fused merge per tile; `download()` brings the results
    back through pinned buffers with asynchronous D2H copies on a side stream (tile k's copy overlaps tile k+1's);
    `assemble()` concatenates the tiles of all ranks on `dst` through the CPU (gloo) group with tensor gathers into a
    preallocated image - no pickling, no per-tile temporaries. There is no GPU<->GPU traffic on this path."""

    def __init__(self, height: int, n_tiles: int, rank: int = 0, world_size: int = 1, median_k: int = 0):
        self.height, self.n_tiles, self.rank, self.world = height, n_tiles, rank, world_size
        self.bounds = row_tile_bounds(height, n_tiles)
        self.mine = tiles_for_rank(n_tiles, rank, world_size)
        self.median_k = median_k
        self.plans = {}
        self._host = {}
        self._copy_stream = None

    def input_rows(self, tile: int) -> Tuple[int, int]:
        r0, r1 = self.bounds[tile]
        return halo_bounds(r0, r1, self.height, self.median_k)

    def add_tile(self, tile: int, frames, exposures, icrf, icrf_diff=None, stds=None, **kw):
        """frames / stds / darks: device tensors covering input_rows(tile); flat / flat_std cover the tile's own rows."""
        from . import engine
        r0, r1 = self.bounds[tile]
        b0, _ = self.input_rows(tile)
        self.plans[tile] = engine.plan_merge(frames, exposures, icrf, icrf_diff, stds, height=self.height, row0=r0, rows=r1 - r0,
                                             buf_row0=b0, **kw)

    def launch(self, stream: Optional[int] = None) -> None:
        for t in self.mine:
            self.plans[t].launch(stream)

    @property
    def algorithmic_bytes(self) -> int:
        return sum(self.plans[t].algorithmic_bytes for t in self.mine)

    def _side_stream(self, dev):
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(dev)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._copy_stream.wait_event(ev)
        return self._copy_stream

    def download(self, into=None):
        """{tile: (val, std | None)} as pinned host tensors; the merges must have been launched on the current stream.
        `into` = (val image, std image | None): pinned whole-image tensors whose rows receive the tiles directly (no per-tile buffers)."""
        if not self.mine:                                   # more ranks than tiles: nothing to bring back
            return {}
        dev = self.plans[self.mine[0]].device
        side = self._side_stream(dev)
        out = {}
        with torch.cuda.stream(side):
            for t in self.mine:
                o = self.plans[t].outputs
                if into is not None:
                    r0, r1 = self.bounds[t]
                    hv = into[0][r0:r1]
                    hs = into[1][r0:r1] if (into[1] is not None and "std" in o) else None
                else:
                    if t not in self._host:                 # (pinned at creation: .pin_memory() on a pageable tensor allocates and copies it)
                        self._host[t] = (torch.empty(o["val"].shape, dtype=torch.float64, pin_memory=True),
                                         torch.empty(o["std"].shape, dtype=torch.float64, pin_memory=True) if "std" in o else None)
                    hv, hs = self._host[t]
                hv.copy_(o["val"], non_blocking=True)
                if hs is not None:
                    hs.copy_(o["std"], non_blocking=True)
                out[t] = (hv, hs)
        side.synchronize()
        return out

    def assemble(self, group=None, dst: int = 0, copy: bool = False):
        """(val, std) of the whole image as host tensors on `dst`, (None, None) elsewhere. The destination rank's own tiles go from the
        device straight into their rows of the (pinned) image; the other ranks' tiles arrive over the CPU group into theirs.
        The returned tensors ARE this object's pinned image buffers: the next assemble() (the next stack of a loop) overwrites them.
        copy=True returns tensors of the caller's own instead (one more host copy of the image)."""
        image = None
        if self.rank == dst and self.mine:
            o = self.plans[self.mine[0]].outputs
            tail = tuple(o["val"].shape[1:])
            key = (self.height,) + tail + ("std" in o,)
            if getattr(self, "_image_key", None) != key:
                self._image = (torch.empty((self.height,) + tail, dtype=torch.float64, pin_memory=True),
                               torch.empty((self.height,) + tail, dtype=torch.float64, pin_memory=True) if "std" in o else None)
                self._image_key = key
            image = self._image
        local = self.download(into=image)
        val, std = gather_tiles(local, self.bounds, group=group, dst=dst, world_size=self.world, rank=self.rank, image=image)
        if copy and val is not None:
            val, std = val.clone(), (None if std is None else std.clone())
        return val, std


def gather_tiles(local: dict, bounds: Sequence[Tuple[int, int]], group=None, dst: int = 0, world_size: int = 1, rank: int = 0, image=None):
    """Assemble {tile index: (val, std | None)} dictionaries of all ranks into one image on `dst`. With one rank this is a
    concatenation into a preallocated buffer; with several, every tile travels as ONE tensor send / receive over the CPU
    (gloo) group straight into its rows of the destination image. `image` = (val, std | None) preallocated on `dst` whose rows already
    hold dst's own tiles (RowTileSet.assemble): those are not copied again - and `image` itself is what is returned on dst (aliased,
    not a copy)."""
    some = next(iter(local.values())) if local else None
    with_std = some is not None and some[1] is not None
    if world_size == 1:
        if some is None:
            raise ValueError("gather_tiles: this (only) rank holds no tiles to assemble")
        shape_tail = tuple(some[0].shape[1:])
        if image is not None:
            return image[0], (image[1] if with_std else None)
        H = bounds[-1][1]
        val = torch.empty((H,) + shape_tail, dtype=torch.float64)
        std = torch.empty((H,) + shape_tail, dtype=torch.float64) if with_std else None
        for t, (v, s) in local.items():
            r0, r1 = bounds[t]
            val[r0:r1] = torch.as_tensor(v)
            if with_std:
                std[r0:r1] = torch.as_tensor(s)
        return val, std
    import torch.distributed as dist
    n_tiles = len(bounds)
    meta = torch.zeros(4, dtype=torch.int64)                      # (W, C, with_std, have_any) agreed through a max-reduce
    if some is not None:
        meta[0], meta[1], meta[2], meta[3] = some[0].shape[1], some[0].shape[2], int(with_std), 1
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    W, Cc, with_std = int(meta[0]), int(meta[1]), bool(meta[2])
    val = std = None
    if rank == dst:
        H = bounds[-1][1]
        if image is not None:
            val, std = image[0], (image[1] if with_std else None)
        else:
            val = torch.empty((H, W, Cc), dtype=torch.float64)
            std = torch.empty((H, W, Cc), dtype=torch.float64) if with_std else None
    reqs = []
    for t in range(n_tiles):
        owner = t % world_size
        r0, r1 = bounds[t]
        if r1 == r0:
            continue
        if owner == dst:
            if rank == dst and image is None:
                val[r0:r1] = torch.as_tensor(local[t][0])
                if with_std:
                    std[r0:r1] = torch.as_tensor(local[t][1])
            continue
        if rank == owner:
            reqs.append(dist.isend(torch.as_tensor(local[t][0]).contiguous(), dst=dst, group=group, tag=2 * t))
            if with_std:
                reqs.append(dist.isend(torch.as_tensor(local[t][1]).contiguous(), dst=dst, group=group, tag=2 * t + 1))
        elif rank == dst:
            reqs.append(dist.irecv(val[r0:r1], src=owner, group=group, tag=2 * t))     # rows of a C-contiguous image: contiguous
            if with_std:
                reqs.append(dist.irecv(std[r0:r1], src=owner, group=group, tag=2 * t + 1))
    for r in reqs:
        r.wait()
    return val, std


def gather_row_tiles(local_val: np.ndarray, local_std: Optional[np.ndarray], group=None, dst: int = 0):
    """Host-side assembly of ONE tile per rank (rank r owns tile r) on rank `dst` through the CPU process group (gloo):
    every tile is received straight into its rows of a preallocated image (tensor send / receive - no pickling).
    Returns (val, std) NumPy arrays on dst, (None, None) elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rows = torch.zeros(world, dtype=torch.int64)
    rows[rank] = local_val.shape[0]
    dist.all_reduce(rows, group=group)
    edges = [0]
    for r in range(world):
        edges.append(edges[-1] + int(rows[r]))
    bounds = [(edges[r], edges[r + 1]) for r in range(world)]
    local = {rank: (torch.as_tensor(np.ascontiguousarray(local_val)),
                    None if local_std is None else torch.as_tensor(np.ascontiguousarray(local_std)))}
    val, std = gather_tiles(local, bounds, group=group, dst=dst, world_size=world, rank=rank)
    if rank != dst:
        return None, None
    return val.numpy(), (None if std is None else std.numpy())
