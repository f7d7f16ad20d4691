"""N > 1 path with the real HIP kernels: three processes (one per rank, gloo group, all on the one GPU of the test box - the
scaling runs give each rank its own GPU) each merge their row tile with the median halo through hm_merge and the tiles are
gathered host-side on rank 0 (camera_linearity_amd/parallel.py): bit-identical to the whole image merged by one process."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hdr_oracle as orc  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    n, h, w = 5, 61, 40
    frames, stds, t = orc.synthetic_stack(31, n, h, w, with_std=True)
    rng = np.random.default_rng(31)
    dark = (rng.random((h, w, 3)) < 0.02).astype(np.uint8) * 200
    dark[0, 0, 0] = dark[h - 1, w - 1, 1] = 255
    flat = rng.integers(160, 240, (h, w, 3)).astype(np.uint8)
    flat_std = np.full((h, w, 3), 0.002)
    icrf, diff = orc.synthetic_icrf()
    return frames, stds, t, dark, flat, flat_std, icrf, diff


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from camera_linearity_amd import parallel
        frames, stds, t, dark, flat, flat_std, icrf, diff = _inputs()
        r0, r1, val, std = parallel.merge_row_tile(frames, t, icrf, diff, stds_host=stds, darks_host=[None, dark, dark, dark, dark],
                                                   dark_min=[256, 13, 13, 13, 13], median_k=3, flat_host=flat, flat_std_host=flat_std,
                                                   ff_mean=[0.78, 0.8, 0.79], ff_std_mean=[0.002] * 3, rank=rank, world_size=world)
        assert (r0, r1) == parallel.row_tile_bounds(frames[0].shape[0], world)[rank]
        gval, gstd = parallel.gather_row_tiles(val, std, dst=0)
        if rank == 0:
            q.put((gval, gstd))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:          # surface the failure to the parent instead of letting it wait for the queue
        q.put(("error", f"rank {rank}: {type(e).__name__}: {e}"))
        raise


def test_row_tiles_three_ranks_on_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from camera_linearity_amd import engine
    frames, stds, t, dark, flat, flat_std, icrf, diff = _inputs()
    dev = torch.device("cuda", 0)
    dk = torch.as_tensor(dark, device=dev)
    whole = engine.merge([torch.as_tensor(f, device=dev) for f in frames], t, icrf, diff, [torch.as_tensor(s, device=dev) for s in stds],
                         darks=[None, dk, dk, dk, dk], dark_min=[256, 13, 13, 13, 13], median_k=3,
                         flat=torch.as_tensor(flat, device=dev), flat_std=torch.as_tensor(flat_std, device=dev),
                         ff_mean=[0.78, 0.8, 0.79], ff_std_mean=[0.002] * 3)
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=150)
    assert got[0] is not None and not (isinstance(got[0], str) and got[0] == "error"), got
    gval, gstd = got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(gval, whole["val"].cpu().numpy())
    assert np.array_equal(gstd, whole["std"].cpu().numpy())


@pytest.mark.parametrize("workload", ["cfg2", "cfg4"])
def test_bench_launches_its_own_ranks(workload):
    """`python bench.py --gpus 2 ...` from a plain shell (no torchrun on the command line): the parent starts its two ranks as a
    child process (here both on the one GPU, gloo carrying the barrier) and rank 0's single JSON line comes through."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo", "--workload", workload,
                          "--steps", "5", "--warmup", "1", "--prewarm-s", "0.05", "--stacks", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["cpu_baseline"] is None
    assert line["ranks"]["ranks_seen"] == 2 and len(line["ranks"]["avg_launch_us_per_rank"]) == 2
    assert line["ranks"]["avg_launch_us_min"] <= line["ranks"]["avg_launch_us_max"]
    assert line["roofline"]["kernel"].startswith("merge_u8_val3")


def _bench(args, timeout=600):
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_PORT"] = str(_free_port())
    return subprocess.run([sys.executable, str(root / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("workload", ["cfg2", "cfg4", "linearity"])
def test_bench_nccl_world_of_one(workload):
    """The RCCL path of bench.py on real hardware, as far as a one-GPU box allows (two nccl ranks cannot share a GPU): a child
    process runs `bench.py --gpus 1 --force-dist --dist-backend nccl`, i.e. init_process_group("nccl", device_id), the
    communicator-creating barrier, the barriers around the timed region and the device-side all-gather of rank_report; cfg4
    also creates its gloo sub-group inside the nccl world and assembles through the shared-memory image."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    out = _bench(["--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--workload", workload, "--steps", "5", "--warmup", "1",
                  "--prewarm-s", "0.05", "--stacks", "1", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks"]["ranks_seen"] == 1
    assert len(line["ranks"]["avg_launch_us_per_rank"]) == 1 and line["ranks"]["avg_launch_us_per_rank"][0] > 0
    if workload == "cfg4":
        assert "shared-memory image" in line["assembly"], line["assembly"]
        assert line["assembly_ms"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N on a node with fewer GPUs and no --share-device: every rank prints ONE line and the run exits non-zero."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count() + 1
    out = _bench(["--gpus", str(n), "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert out.returncode != 0
    assert "needs GPU" in out.stderr and "--share-device" in out.stderr, out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def _worker_tileset(rank, world, port, q, shared):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        from camera_linearity_amd import parallel
        frames, stds, t, dark, flat, flat_std, icrf, diff = _inputs()
        H = frames[0].shape[0]
        n_tiles = 5
        tiles = parallel.RowTileSet(H, n_tiles, rank, world, median_k=3)
        up = lambda a, lo, hi: torch.as_tensor(np.ascontiguousarray(a[lo:hi]), device=dev)   # noqa: E731
        for tile in tiles.mine:
            r0, r1 = tiles.bounds[tile]
            b0, b1 = tiles.input_rows(tile)
            dk = up(dark, b0, b1)
            tiles.add_tile(tile, [up(f, b0, b1) for f in frames], t, icrf, diff, [up(s, b0, b1) for s in stds],
                           darks=[None, dk, dk, dk, dk], dark_min=[256, 13, 13, 13, 13], median_k=3, flat=up(flat, r0, r1),
                           flat_std=up(flat_std, r0, r1), ff_mean=[0.78, 0.8, 0.79], ff_std_mean=[0.002] * 3)
        tiles.launch()
        for _ in range(2):                       # the second call reuses the mapped image
            val, std = tiles.assemble(dst=0, shared=shared)
        if rank == 0:
            q.put((val.numpy().copy(), std.numpy().copy(), tiles.assembly_path))
        del val, std
        dist.barrier()
        tiles.close()
        dist.destroy_process_group()
    except Exception as e:
        q.put(("error", f"rank {rank}: {type(e).__name__}: {e}", ""))
        raise


@pytest.mark.parametrize("shared", [True, False])
def test_row_tile_set_two_ranks_assembly(shared):
    """RowTileSet.assemble with two ranks on the one GPU: five tiles (median halo, flat field, std) land in ONE shared-memory image
    that both ranks page-lock and copy into (shared=True), or travel as tensor sends (shared=False, the fallback): both
    bit-identical to the whole image merged by one process."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    from camera_linearity_amd import engine
    frames, stds, t, dark, flat, flat_std, icrf, diff = _inputs()
    dev = torch.device("cuda", 0)
    dk = torch.as_tensor(dark, device=dev)
    whole = engine.merge([torch.as_tensor(f, device=dev) for f in frames], t, icrf, diff, [torch.as_tensor(s, device=dev) for s in stds],
                         darks=[None, dk, dk, dk, dk], dark_min=[256, 13, 13, 13, 13], median_k=3,
                         flat=torch.as_tensor(flat, device=dev), flat_std=torch.as_tensor(flat_std, device=dev),
                         ff_mean=[0.78, 0.8, 0.79], ff_std_mean=[0.002] * 3)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tileset, args=(r, world, port, q, shared)) for r in range(world)]
    for p in procs:
        p.start()
    gval, gstd, path = q.get(timeout=150)
    assert not (isinstance(gval, str) and gval == "error"), gstd
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(gval, whole["val"].cpu().numpy())
    assert np.array_equal(gstd, whole["std"].cpu().numpy())
    assert path == ("shared-memory image" if shared else "tensor sends over the CPU group"), path


def test_example_one_image_on_n_gpus_same_image_for_any_rank_count():
    """examples/merge_one_image_on_n_gpus.py (the user-level recipe: RowTileSet + a gloo group, no collective on the data path) with one
    process and with three ranks sharing the GPU: the assembled image has the same mean to the last printed digit, the three-rank run
    went through the shared-memory image."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pathlib
    import re
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parent.parent
    ex = str(root / "examples" / "merge_one_image_on_n_gpus.py")
    size = ["--frames", "5", "--height", "1024", "--width", "640", "--tiles", "8", "--std"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    one = subprocess.run([sys.executable, ex] + size, capture_output=True, text=True, timeout=600, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), ex, "--share-device"] + size, capture_output=True, text=True, timeout=600, env=env)
    assert three.returncode == 0, three.stderr[-2000:]
    pat = re.compile(r"val \((\d+), (\d+), 3\) mean ([0-9.]+), std mean ([0-9.e+-]+)")
    m1, m3 = pat.search(one.stdout), pat.search(three.stdout)
    assert m1 and m3, (one.stdout, three.stdout)
    assert m1.groups() == m3.groups() and m1.group(1) == "1024"
    assert "on 3 rank(s)" in three.stdout and "shared-memory image" in three.stdout and "pinned image" in one.stdout
