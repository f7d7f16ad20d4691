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
