"""N > 1 path on CPU: two gloo ranks each merge their row tile (+ median halo) and the tiles are
gathered host-side on rank 0 (camera_linearity_amd/parallel.py). There is no GPU here and the product has
no CPU fallback, so each rank's tile is computed with the ORACLE; what is under test is the sharding
itself - tile bounds, halo rows, 'reflect' only at true image edges, gather order - which must
reproduce the whole-image oracle result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hdr_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stack():
    n, h, w = 4, 23, 10
    frames, stds, t = orc.synthetic_stack(21, n, h, w, with_std=True)
    rng = np.random.default_rng(21)
    dark = rng.integers(0, 30, size=(h, w, 3)).astype(np.uint8)
    dark[0, 0, 0] = dark[h - 1, w - 1, 1] = dark[11, 4, 2] = dark[12, 4, 2] = 255
    icrf, diff = orc.synthetic_icrf()
    return frames, stds, t, dark, icrf, diff


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from camera_linearity_amd import parallel
    frames, stds, t, dark, icrf, diff = _stack()
    H = frames[0].shape[0]
    r0, r1 = parallel.row_tile_bounds(H, world)[rank]
    b0, b1 = parallel.halo_bounds(r0, r1, H, 3)
    # the tile sees only its rows + halo; emulate 'reflect at true edges only' by merging the halo'd band
    # and cropping - valid because the band contains every row the tile's medians touch
    band = slice(b0, b1)
    darkv = orc.unit_from_u8(dark[band])
    # bands that do not start/end at an image edge must not reflect there: pad with the real neighbours (already in band)
    out = orc.merge([f[band] for f in frames], t, icrf, diff, stds=[s[band] for s in stds],
                    darks=[None, darkv, darkv, darkv], dark_threshold=0.075, median_k=3)
    val = out["val"][r0 - b0:r1 - b0]
    std = out["std"][r0 - b0:r1 - b0]
    gval, gstd = parallel.gather_row_tiles(val, std, dst=0)
    if rank == 0:
        q.put((gval, gstd))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_tiles_over_gloo(world):
    frames, stds, t, dark, icrf, diff = _stack()
    darkv = orc.unit_from_u8(dark)
    whole = orc.merge(frames, t, icrf, diff, stds=stds, darks=[None, darkv, darkv, darkv], dark_threshold=0.075, median_k=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    gval, gstd = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(gval, whole["val"])
    assert np.array_equal(gstd, whole["std"])


def _worker_tiles(rank, world, port, q, n_tiles):
    """Several tiles per rank (config 4's shape: 8 tiles over G ranks), assembled with parallel.gather_tiles."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from camera_linearity_amd import parallel
    frames, stds, t, dark, icrf, diff = _stack()
    H = frames[0].shape[0]
    bounds = parallel.row_tile_bounds(H, n_tiles)
    darkv = orc.unit_from_u8(dark)
    local = {}
    for tile in parallel.tiles_for_rank(n_tiles, rank, world):
        r0, r1 = bounds[tile]
        b0, b1 = parallel.halo_bounds(r0, r1, H, 3)
        band = slice(b0, b1)
        out = orc.merge([f[band] for f in frames], t, icrf, diff, stds=[s[band] for s in stds],
                        darks=[None, darkv[band], darkv[band], darkv[band]], dark_threshold=0.075, median_k=3)
        local[tile] = (torch.as_tensor(out["val"][r0 - b0:r1 - b0].copy()), torch.as_tensor(out["std"][r0 - b0:r1 - b0].copy()))
    val, std = parallel.gather_tiles(local, bounds, dst=0, world_size=world, rank=rank)
    if rank == 0:
        q.put((val.numpy(), std.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tiles", [(2, 5), (3, 8)])
def test_many_row_tiles_per_rank_over_gloo(world, n_tiles):
    frames, stds, t, dark, icrf, diff = _stack()
    darkv = orc.unit_from_u8(dark)
    whole = orc.merge(frames, t, icrf, diff, stds=stds, darks=[None, darkv, darkv, darkv], dark_threshold=0.075, median_k=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tiles, args=(r, world, port, q, n_tiles)) for r in range(world)]
    for p in procs:
        p.start()
    gval, gstd = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(gval, whole["val"])
    assert np.array_equal(gstd, whole["std"])


def test_gather_tiles_single_rank():
    from camera_linearity_amd import parallel
    rng = np.random.default_rng(3)
    img = rng.random((17, 5, 3))
    bounds = parallel.row_tile_bounds(17, 4)
    local = {t: (torch.as_tensor(img[r0:r1].copy()), None) for t, (r0, r1) in enumerate(bounds)}
    val, std = parallel.gather_tiles(local, bounds)
    assert std is None and np.array_equal(val.numpy(), img)


def _worker_shared(rank, world, port, q, shm_dir):
    """The shared-memory assembly of RowTileSet.assemble without a GPU: every rank writes its tiles' rows straight into the one
    image (here with a host copy where the product issues a D2H copy); the group carries a name, a flag and a barrier."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from camera_linearity_amd import parallel
    if shm_dir is not None:
        parallel.SharedHostImage.DIR = shm_dir
    rng = np.random.default_rng(5)
    H, W, n_tiles = 37, 6, 5
    img = rng.random((H, W, 3))
    sd = rng.random((H, W, 3))
    bounds = parallel.row_tile_bounds(H, n_tiles)
    shared = parallel.SharedHostImage.open((H, W, 3), True, group=None, rank=rank, dst=0, world_size=world)
    if shared is None:
        q.put(("fallback", rank))
    else:
        assert not shared.pin()                       # no GPU here: nothing to page-lock for, and nothing raised
        for t in parallel.tiles_for_rank(n_tiles, rank, world):
            r0, r1 = bounds[t]
            shared.val[r0:r1].copy_(torch.as_tensor(img[r0:r1]))
            shared.std[r0:r1].copy_(torch.as_tensor(sd[r0:r1]))
        dist.barrier()
        if rank == 0:
            left = [f for f in os.listdir(parallel.SharedHostImage.DIR) if f == shared.name]
            q.put((shared.val.numpy().copy(), shared.std.numpy().copy(), left))
        dist.barrier()
        shared.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shared_host_image_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared, args=(r, world, port, q, None)) for r in range(world)]
    for p in procs:
        p.start()
    val, std, left = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    assert np.array_equal(val, rng.random((37, 6, 3)))
    assert np.array_equal(std, rng.random((37, 6, 3)))
    assert left == []                                  # the name is unlinked as soon as every rank is attached


def test_shared_host_image_falls_back_on_every_rank(tmp_path):
    """No usable shared-memory directory on the creating rank: open() returns None on EVERY rank (the caller then takes gather_tiles)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared, args=(r, 2, port, q, str(tmp_path / "absent"))) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [("fallback", 0), ("fallback", 1)]


def test_shared_host_image_single_process():
    from camera_linearity_amd import parallel
    s = parallel.SharedHostImage.open((4, 3, 3), False)
    assert s is not None and s.std is None and s.val.shape == (4, 3, 3)
    s.val.fill_(2.5)
    assert float(s.val.sum()) == 90.0
    assert not os.path.exists(os.path.join(parallel.SharedHostImage.DIR, s.name))
    s.close()
    s.close()


def test_tiles_of_an_odd_width_image_start_at_even_element_offsets():
    """W * C odd (e.g. 333 x 3): the streaming kernels read element pairs, so a tile's first element must sit an even number of elements
    into its input buffer. With row_elems given, tiles start on even rows and take an even number of halo rows above - covering, disjoint,
    the halo still at least floor(k / 2) rows on both sides."""
    from camera_linearity_amd import parallel as P
    for height, tiles, k, row_elems in ((1001, 8, 3, 999), (1001, 8, 5, 999), (64, 7, 7, 333), (10, 3, 3, 21), (1001, 8, 3, 1000), (5, 8, 3, 9)):
        bounds = P.row_tile_bounds(height, tiles, row_elems)
        assert bounds[0][0] == 0 and bounds[-1][1] == height and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
        for r0, r1 in bounds:
            b0, b1 = P.halo_bounds(r0, r1, height, k, row_elems)
            assert b0 <= max(0, r0 - k // 2) and b1 == min(height, r1 + k // 2) and b0 >= 0
            if row_elems % 2 == 1:
                assert r0 % 2 == 0 and ((r0 - b0) * row_elems) % 2 == 0, (height, tiles, k, r0, b0)
    ts = P.RowTileSet(1001, 8, median_k=3, row_elems=999)
    assert ts.input_rows(1) == (ts.bounds[1][0] - 2, ts.bounds[1][1] + 1)
    assert P.RowTileSet(1001, 8, median_k=3).input_rows(1) == (124, 251)          # unchanged without row_elems
