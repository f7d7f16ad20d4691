#!/usr/bin/env python3
"""TIFF read throughput (host only): 4096 x 4096 x 3 uint8 images - uncompressed, and LZW as OpenCV / libtiff write it (via PIL when it is
installed) for a smooth, a photo-like (gradient + sensor noise) and a noise image; float64 uncompressed. Pixel MB/s, second read of each file."""
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from camera_linearity_amd import tiff_io as T  # noqa: E402

d = pathlib.Path(tempfile.mkdtemp())
rng = np.random.default_rng(0)
ramp = np.add.outer(np.arange(4096), np.arange(4096))[:, :, None]
images = {"smooth": (ramp // 37 % 256 * np.ones(3)).astype(np.uint8),
          "photo-like": np.clip(ramp / 40 + rng.normal(size=(4096, 4096, 3)) * 2, 0, 255).astype(np.uint8),
          "noise": (rng.random((4096, 4096, 3)) * 255).astype(np.uint8)}


def timed_read(path, flag=None):
    T.imread(path) if flag is None else T.imread(path, flag)
    t0 = time.perf_counter()
    a = T.imread(path) if flag is None else T.imread(path, flag)
    return a, time.perf_counter() - t0


for name, img in images.items():
    T.imwrite(d / "raw.tif", img)
    a, dt = timed_read(d / "raw.tif")
    line = f"{name}: uncompressed {img.nbytes / dt / 1e6:.0f} MB/s"
    try:
        from PIL import Image
        Image.fromarray(img[:, :, ::-1]).save(d / "lzw.tif", compression="tiff_lzw")
        a, dt = timed_read(d / "lzw.tif")
        assert np.array_equal(a, img)
        line += f", LZW ({(d / 'lzw.tif').stat().st_size / 1e6:.1f} MB file) {img.nbytes / dt / 1e6:.0f} MB/s"
    except ImportError:
        pass
    print(line, flush=True)
f64 = rng.random((2048, 2048, 3))
T.imwrite(d / "f64.tif", f64)
a, dt = timed_read(d / "f64.tif", T.IMREAD_UNCHANGED)
print(f"float64 2048 x 2048 x 3 uncompressed: {f64.nbytes / dt / 1e6:.0f} MB/s")
import shutil  # noqa: E402
shutil.rmtree(d)
