"""Data-parallel sharding of the merge across the GPUs of one node.

The merge is per-pixel independent (modules/exposure_series.py:388-389), so there is NO data-path
collective: one process per GPU (torch.distributed, launched with torch.distributed.run), each rank
merges its own units and results are assembled host-side. Two shardings:

  * row tiles   - one large stack split into contiguous row ranges (HWC row-major => one contiguous
                  byte range per frame). The hot-pixel median needs a floor(k/2)-row halo of INPUT rows
                  from the neighbouring tile; 'reflect' applies only at the true image edges. The two
                  flat-field ROI means are global scalars, computed once and passed by value.
  * whole stacks - a batch of independent stacks dealt round-robin to the ranks.

Result assembly is host-side: the ranks of a node map ONE image in POSIX shared memory (SharedHostImage), page-lock it
and copy their tiles from the device straight into their rows of it; the CPU (gloo) group carries the segment's name, a
6-number shape agreement and a barrier - no pixel travels through it. (gather_tiles, tensor sends over the CPU group, stays as
the fallback for when /dev/shm cannot hold the image, and for tiles that are host arrays already.)
"""
from __future__ import annotations

import mmap
import os
import secrets
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def row_tile_bounds(height: int, world_size: int, row_elems: Optional[int] = None) -> List[Tuple[int, int]]:
    """Rank r owns rows [r*H/G, (r+1)*H/G) (integer floor): contiguous, disjoint, covering, balanced to 1 row.
    row_elems = W * C: when it is odd, tiles start on EVEN rows (balanced to 2 rows), so that a tile and its halo can always start an even
    number of elements into the image (halo_bounds)."""
    if height < 0 or world_size < 1:
        raise ValueError("height >= 0 and world_size >= 1 required")
    cut = [(r * height) // world_size for r in range(world_size + 1)]
    if row_elems is not None and row_elems % 2 == 1:
        cut = [c & ~1 for c in cut[:-1]] + [height]
    return [(cut[r], cut[r + 1]) for r in range(world_size)]


def halo_bounds(row0: int, row1: int, height: int, median_k: int = 0, row_elems: Optional[int] = None) -> Tuple[int, int]:
    """Input rows a tile must hold: its own rows plus floor(k/2) rows either side, clipped to the image.
    row_elems = W * C of the image: when it is odd, the rows above the tile are made an EVEN number (one more row, if the image has
    it) - the streaming kernels read element pairs and need the tile's first element, (row0 - buffer row0) * W * C elements into the
    buffer, at an even offset; an odd one sends the whole tile through the one-element-per-thread generic kernel (same bits, 3-4 x slower)."""
    r = median_k // 2 if median_k else 0
    b0, b1 = max(0, row0 - r), min(height, row1 + r)
    if row_elems is not None and row_elems % 2 == 1 and (row0 - b0) % 2 == 1 and b0 > 0:
        b0 -= 1
    return b0, b1


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
    row_elems = int(np.prod(frames_host[0].shape[1:], dtype=np.int64))           # W * C: odd -> even tile starts and halo (halo_bounds)
    r0, r1 = row_tile_bounds(H, world_size, row_elems)[rank] if tile is None else tile
    use_hot = darks_host is not None and any(d is not None for d in darks_host)
    b0, b1 = halo_bounds(r0, r1, H, median_k if use_hot else 0, row_elems)
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


class SharedHostImage:
    """ONE (H, W, C) float64 image - and optionally a second one for std - in POSIX shared memory (/dev/shm), mapped by every
    rank of the node. Rank `dst` creates the segment, the others attach to it by name; every rank page-locks its mapping
    (hipHostRegister through torch's cudart binding) so that its D2H copies land in the image directly and asynchronously, with
    no staging buffer and no inter-process send. Construction is COLLECTIVE over `group` (a CPU / gloo group, or the default
    group): the name travels by broadcast_object_list and a MIN-reduce agrees that every rank mapped it - if any rank could not
    (no /dev/shm, not enough room in it), open() returns None on ALL ranks and the caller falls back to gather_tiles. The
    creator unlinks the name as soon as everyone is attached, so nothing is left in /dev/shm if a rank dies later (the pages
    live until the last mapping goes)."""

    DIR = "/dev/shm"

    def __init__(self, mm, shape, with_std: bool, name: str, owner: bool):
        self._mm, self.shape, self.name, self.owner = mm, tuple(shape), name, owner
        n = int(np.prod(shape, dtype=np.int64))
        self.val = torch.frombuffer(mm, dtype=torch.float64, count=n, offset=0).reshape(self.shape)
        self.std = torch.frombuffer(mm, dtype=torch.float64, count=n, offset=8 * n).reshape(self.shape) if with_std else None
        self.nbytes = 8 * n * (2 if with_std else 1)
        self.pinned = False

    @staticmethod
    def _map(path: str, nbytes: int, create: bool):
        fd = os.open(path, os.O_RDWR | (os.O_CREAT | os.O_EXCL if create else 0), 0o600)
        try:
            if create:
                os.posix_fallocate(fd, 0, nbytes)           # ENOSPC here, not a SIGBUS at the first touch of a page tmpfs cannot back
            return mmap.mmap(fd, nbytes)
        except BaseException:
            if create:
                os.unlink(path)
            raise
        finally:
            os.close(fd)

    @classmethod
    def open(cls, shape, with_std: bool, group=None, rank: int = 0, dst: int = 0, world_size: int = 1) -> Optional["SharedHostImage"]:
        import torch.distributed as dist
        nbytes = 8 * int(np.prod(shape, dtype=np.int64)) * (2 if with_std else 1)
        collective = world_size > 1 or (group is not None)
        if collective and group is None and dist.get_backend() == "nccl":
            raise ValueError("SharedHostImage.open exchanges host tensors: pass a CPU process group (dist.new_group(backend='gloo')) "
                             "when the default group is nccl")
        name, mm = None, None
        if rank == dst and nbytes > 0:
            name = f"hm_image_{os.getpid()}_{secrets.token_hex(6)}"
            try:
                mm = cls._map(os.path.join(cls.DIR, name), nbytes, create=True)
            except OSError:
                name, mm = None, None
        if collective:
            box = [name]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
            name = box[0]
            if rank != dst and name is not None:
                try:
                    mm = cls._map(os.path.join(cls.DIR, name), nbytes, create=False)
                except OSError:
                    mm = None
            ok = torch.tensor([1 if mm is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)               # also the "everyone is attached" barrier
            all_ok = bool(ok.item())
        else:
            all_ok = mm is not None
        if rank == dst and name is not None:
            try:
                os.unlink(os.path.join(cls.DIR, name))
            except OSError:
                pass
        if not all_ok:
            if mm is not None:
                mm.close()
            return None
        return cls(mm, shape, with_std, name, rank == dst)

    def pin(self) -> bool:
        """Page-lock this process's mapping for the GPU (every rank pins its own mapping). False = the runtime refused: copies into
        the image still work, synchronously through the runtime's own staging."""
        if self.pinned or not torch.cuda.is_available():
            return self.pinned
        rc = torch.cuda.cudart().cudaHostRegister(self.val.data_ptr(), self.nbytes, 0)
        self.pinned = int(rc) == 0
        if not self.pinned:                                  # do not leave the refusal behind as the runtime's sticky "last error"
            try:
                import ctypes
                ctypes.CDLL("libamdhip64.so").hipGetLastError()
            except OSError:
                pass
        return self.pinned

    def close(self) -> None:
        if self._mm is None:
            return
        if self.pinned:
            torch.cuda.cudart().cudaHostUnregister(self.val.data_ptr())
            self.pinned = False
        self.val = self.std = None
        try:
            self._mm.close()
        except BufferError:          # a caller still holds a view of the image: the mapping goes when that view does
            pass
        self._mm = None


class RowTileSet:
    """The row tiles of ONE large image that this rank owns, resident on its GPU (config 4: a 15 x 8192 x 8192 x 3
    stack cut into 8 tiles of 1024 rows). `launch()` enqueues one fused merge per tile; `download()` brings the results
    back through pinned buffers with asynchronous D2H copies on a side stream (tile k's copy overlaps tile k+1's);
    `assemble()` puts the tiles of all ranks into ONE host image on `dst`: a shared-memory image every rank copies its own rows
    into (SharedHostImage), or - fallback - tensor sends over the CPU (gloo) group. There is no GPU<->GPU traffic on this path."""

    def __init__(self, height: int, n_tiles: int, rank: int = 0, world_size: int = 1, median_k: int = 0, row_elems: Optional[int] = None):
        """row_elems = W * C (optional): lets input_rows() keep a tile's first element at an even offset for odd W * C (halo_bounds)."""
        self.row_elems = row_elems
        self.height, self.n_tiles, self.rank, self.world = height, n_tiles, rank, world_size
        self.bounds = row_tile_bounds(height, n_tiles, row_elems)
        self.mine = tiles_for_rank(n_tiles, rank, world_size)
        self.median_k = median_k
        self.plans = {}
        self._host = {}
        self._copy_stream = None
        self.assembly_path = None

    def input_rows(self, tile: int) -> Tuple[int, int]:
        r0, r1 = self.bounds[tile]
        return halo_bounds(r0, r1, self.height, self.median_k, self.row_elems)

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

    def _agree_shape(self, group):
        """(W, C, with_std) of the image, the same on every rank (ranks that own no tile learn it from the others)."""
        import torch.distributed as dist
        meta = torch.zeros(3, dtype=torch.int64)
        if self.mine:
            o = self.plans[self.mine[0]].outputs
            meta[0], meta[1], meta[2] = o["val"].shape[1], o["val"].shape[2], int("std" in o)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
        return int(meta[0]), int(meta[1]), bool(meta[2])

    def assemble(self, group=None, dst: int = 0, copy: bool = False, shared: bool = True):
        """(val, std) of the whole image as host tensors on `dst`, (None, None) elsewhere.
        One rank and no group: the tiles go from the device straight into their rows of a pinned image of this object's.
        Several ranks (or a group): the image lives in POSIX shared memory that every rank maps and page-locks
        (SharedHostImage); every rank copies its tiles from its GPU straight into its rows of it and a barrier on the CPU group
        says "all rows are there" - nothing is sent between processes. shared=False, or a /dev/shm that cannot hold the image,
        selects the older path: per-tile tensor sends over the CPU group into dst's pinned image (gather_tiles).
        `self.assembly_path` names the path the last call took. The returned tensors ARE this object's image buffers: the next
        assemble() (the next stack of a loop) overwrites them. copy=True returns tensors of the caller's own instead."""
        collective = self.world > 1 or group is not None
        if collective:
            import torch.distributed as dist
            if group is None and dist.get_backend() == "nccl":
                raise ValueError("RowTileSet.assemble exchanges host tensors: pass a CPU process group (dist.new_group(backend='gloo')) "
                                 "when the default group is nccl")
        if collective and shared:
            W, Cc, with_std = self._agree_shape(group)
            key = (self.height, W, Cc, with_std)
            if getattr(self, "_shared_key", None) != key:
                if getattr(self, "_shared", None) is not None:
                    self._shared.close()
                self._shared = SharedHostImage.open((self.height, W, Cc), with_std, group=group, rank=self.rank, dst=dst,
                                                    world_size=self.world)
                self._shared_key = key
                if self._shared is not None:
                    self._shared.pin()
            img = self._shared
            if img is not None:
                self.download(into=(img.val, img.std))
                dist.barrier(group=group)                  # every rank's rows have landed
                self.assembly_path = "shared-memory image" + ("" if img.pinned else " (not page-locked)")
                if self.rank != dst:
                    return None, None
                val, std = img.val, img.std
                if copy:
                    val, std = val.clone(), (None if std is None else std.clone())
                return val, std
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
        self.assembly_path = "pinned image" if not collective else "tensor sends over the CPU group"
        val, std = gather_tiles(local, self.bounds, group=group, dst=dst, world_size=self.world, rank=self.rank, image=image)
        if copy and val is not None:
            val, std = val.clone(), (None if std is None else std.clone())
        return val, std

    def close(self) -> None:
        if getattr(self, "_shared", None) is not None:
            self._shared.close()
            self._shared, self._shared_key = None, None


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
