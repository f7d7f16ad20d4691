"""TIFF reading / writing with OpenCV's conventions, without OpenCV.

The reference moves every image through `cv.imread` / `cv.imwrite` (modules/image_set.py:214-243, 264-363;
modules/video_processing.py:236): 8-bit BGR TIFFs for acquired frames, float64 three-channel TIFFs for the
` STD.tif` / ` HDR.tif` companions. OpenCV is not available here, so this module is a small codec for exactly
that family of files:

  read   classic and BigTIFF, either byte order, strips (not tiles), chunky planar configuration,
         1/3/4 samples of uint8 / uint16 / float32 / float64, Compression none (1), LZW (5, with libtiff's early
         change), Deflate (8 / 32946), PackBits (32773), Predictor 1 / 2 (horizontal differencing)
  write  classic TIFF (BigTIFF when the file would pass 4 GiB), one strip per ~8 KiB of rows like OpenCV,
         uncompressed, uint8 / uint16 / float32 / float64

`imread` / `imwrite` follow OpenCV's channel convention: files hold RGB(A), arrays are BGR(A). OpenCV applies
the swap on both sides for every depth, so a file written here and read by the reference (or the reverse) shows
the same channel order.  `imread(path)` (no flag) returns 8-bit BGR like `cv.imread(path)`,
`imread(path, IMREAD_UNCHANGED)` returns the stored dtype like `cv.imread(path, cv.IMREAD_UNCHANGED)`.

Parity note: the reference holds no TIFF fixtures and cv2 is absent, so interoperability is checked against
Pillow/libtiff (tests/test_tiff_io.py) for the integer formats; the float64 three-channel layout (SampleFormat 3,
BitsPerSample 64,64,64, RGB order in the file) follows the TIFF 6.0 specification and is parity-unpinned against cv2.

The byte-serial LZW / PackBits loops run in libhdrmerge.so's host code (hm_tiff_lzw_decode); strips are decoded
on a thread pool.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Optional

import numpy as np

IMREAD_UNCHANGED = -1          # cv.IMREAD_UNCHANGED
IMREAD_COLOR = 1               # cv.IMREAD_COLOR (the default of cv.imread)

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 6: "b", 7: "B", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}


class TiffError(ValueError):
    pass


def _native():
    from . import _native as nat          # deferred: only compressed strips need the library
    return nat


def _read_ifd(buf: memoryview, bo: str, big: bool, off: int):
    tags = {}
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, inline = off + 8, 20, 8
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, inline = off + 2, 12, 4
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (cnt,) = struct.unpack_from(bo + ("Q" if big else "I"), buf, e + 4)
        voff = e + (12 if big else 8)
        size = _TYPE_SIZES.get(typ)
        if size is None:
            continue
        if size * cnt > len(buf):
            raise TiffError(f"TIFF tag {tag} claims {cnt} values, more than the file holds")
        if size * cnt > inline:
            (voff,) = struct.unpack_from(bo + ("Q" if big else "I"), buf, voff)
        if typ in (5, 10):
            raw = struct.unpack_from(bo + ("I" if typ == 5 else "i") * (2 * cnt), buf, voff)
            tags[tag] = tuple(raw[2 * k] / raw[2 * k + 1] if raw[2 * k + 1] else 0.0 for k in range(cnt))
        elif typ == 2:
            tags[tag] = bytes(buf[voff:voff + cnt])
        else:
            tags[tag] = struct.unpack_from(bo + _TYPE_FMT[typ] * cnt, buf, voff)
    return tags


def _decode_strip(data: bytes, compression: int, expected: int) -> bytes:
    if compression == 1:
        return data
    if compression in (8, 32946):
        return zlib.decompress(data)
    if compression in (5, 32773):
        nat = _native()
        out = (C.c_uint8 * expected)()
        fn = nat.lib.hm_tiff_lzw_decode if compression == 5 else nat.lib.hm_tiff_packbits_decode
        n = fn(data, len(data), out, expected)
        if n < 0:
            raise TiffError(f"corrupt {'LZW' if compression == 5 else 'PackBits'} strip ({nat.strerror(int(n))})")
        return bytes(memoryview(out)[:n])
    raise NotImplementedError(f"TIFF compression {compression} is not supported")


def read_tiff(path) -> np.ndarray:
    """First image of a TIFF file as stored: (H, W) or (H, W, S) in FILE sample order (RGB), file dtype.
    Malformed files raise TiffError (a ValueError)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    try:
        return _read_tiff(buf)
    except (struct.error, IndexError, zlib.error, OverflowError, MemoryError) as e:
        raise TiffError(f"malformed TIFF file {path}: {e}") from e


def _read_tiff(buf: memoryview) -> np.ndarray:
    if len(buf) < 8:
        raise TiffError("not a TIFF file (too short)")
    bo = {b"II": "<", b"MM": ">"}.get(bytes(buf[:2]))
    if bo is None:
        raise TiffError("not a TIFF file (byte-order mark)")
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 42:
        big = False
        (ifd,) = struct.unpack_from(bo + "I", buf, 4)
    elif magic == 43:
        big = True
        (ifd,) = struct.unpack_from(bo + "Q", buf, 8)
    else:
        raise TiffError("not a TIFF file (magic)")
    t = _read_ifd(buf, bo, big, ifd)
    try:
        W, H = int(t[256][0]), int(t[257][0])
    except KeyError as e:
        raise TiffError("TIFF without ImageWidth / ImageLength") from e
    if 322 in t or 324 in t:
        raise NotImplementedError("tiled TIFF files are not supported (OpenCV and the reference write strips)")
    spp = int(t.get(277, (1,))[0])
    bits = t.get(258, (1,))
    if len(set(bits)) != 1:
        raise NotImplementedError(f"mixed BitsPerSample {bits}")
    bps = int(bits[0])
    fmt = int(t.get(339, (1,))[0])
    comp = int(t.get(259, (1,))[0])
    predictor = int(t.get(317, (1,))[0])
    if int(t.get(284, (1,))[0]) != 1 and spp > 1:
        raise NotImplementedError("planar (separate) sample layout is not supported")
    kind = {(1, 8): "u1", (1, 16): "u2", (3, 32): "f4", (3, 64): "f8", (2, 8): "i1", (2, 16): "i2", (1, 32): "u4"}.get((fmt, bps))
    if kind is None:
        raise NotImplementedError(f"SampleFormat {fmt} with {bps} bits per sample")
    dtype = np.dtype(bo + kind) if kind[1] != "1" else np.dtype(kind)
    rps = int(t.get(278, (H,))[0])
    rps = min(rps, H) if rps > 0 else H
    offsets, counts = t.get(273), t.get(279)
    if offsets is None:
        raise TiffError("TIFF without StripOffsets")
    n_strips = (H + rps - 1) // rps
    if len(offsets) < n_strips:
        raise TiffError("StripOffsets shorter than the number of strips")
    row_bytes = W * spp * dtype.itemsize
    if counts is None:
        if comp != 1:
            raise TiffError("compressed TIFF without StripByteCounts")
        counts = [row_bytes * min(rps, H - s * rps) for s in range(n_strips)]
    out = np.empty((H, W * spp), dtype=dtype)
    out_bytes = out.view(np.uint8).reshape(H, row_bytes)

    def one(s: int):
        rows = min(rps, H - s * rps)
        want = rows * row_bytes
        o, c = int(offsets[s]), int(counts[s])
        if o + c > len(buf):
            raise TiffError("strip beyond the end of the file")
        data = _decode_strip(bytes(buf[o:o + c]) if comp != 1 else buf[o:o + c], comp, want)
        if len(data) < want:
            raise TiffError(f"strip {s} decodes to {len(data)} bytes, expected {want}")
        out_bytes[s * rps:s * rps + rows] = np.frombuffer(data, dtype=np.uint8, count=want).reshape(rows, row_bytes)

    if comp == 1 or n_strips == 1:
        for s in range(n_strips):
            one(s)
    else:
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:
            list(pool.map(one, range(n_strips)))
    img = out.reshape(H, W, spp)
    if predictor == 2:
        if dtype.kind not in "ui":
            raise NotImplementedError("horizontal predictor on floating-point samples")
        native = img.astype(dtype.newbyteorder("="), copy=False)
        img = np.cumsum(native, axis=1, dtype=native.dtype)             # modular, like the encoder's differences
    elif predictor != 1:
        raise NotImplementedError(f"TIFF predictor {predictor}")
    img = np.ascontiguousarray(img.astype(dtype.newbyteorder("="), copy=False))
    photometric = int(t.get(262, (1,))[0])
    if photometric == 0 and spp == 1 and dtype.kind == "u":               # WhiteIsZero
        img = np.iinfo(img.dtype).max - img
    return img[:, :, 0] if spp == 1 else img


def _swap_rb(a: np.ndarray) -> np.ndarray:
    if a.ndim == 3 and a.shape[2] in (3, 4):
        out = np.empty_like(a)                      # plane by plane: 1.7 x faster than a copy + fancy-index swap on a 4096 x 4096 x 3 image
        out[..., 0], out[..., 1], out[..., 2] = a[..., 2], a[..., 1], a[..., 0]
        if a.shape[2] == 4:
            out[..., 3] = a[..., 3]
        return out
    return a


def imread(path, flags: int = IMREAD_COLOR) -> Optional[np.ndarray]:
    """cv.imread for TIFF files. Returns None when the file does not exist (OpenCV's behaviour, which
    ImageSet.load_std_image relies on, modules/image_set.py:237-239).
    IMREAD_UNCHANGED: stored dtype, BGR(A) order, 2-D for one sample. Default: 3-channel 8-bit BGR."""
    path = Path(path)
    if not path.exists():
        return None
    img = _swap_rb(read_tiff(path))
    if flags == IMREAD_UNCHANGED:
        return img
    if img.dtype == np.uint16:
        img = (img >> 8).astype(np.uint8)
    elif img.dtype.kind == "f":
        img = np.clip(np.around(img * 255.0), 0, 255).astype(np.uint8)
    elif img.dtype != np.uint8:
        raise NotImplementedError(f"8-bit conversion of {img.dtype} samples")
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    elif img.shape[2] == 4:
        img = img[:, :, :3]
    return np.ascontiguousarray(img)


def imwrite(path, img) -> bool:
    """cv.imwrite for TIFF files: (H, W) or (H, W, 3|4) arrays in BGR(A) order; uint8 / uint16 / float32 / float64.
    Uncompressed strips, RGB(A) order in the file."""
    a = np.asarray(img)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype not in (np.uint8, np.uint16, np.float32, np.float64):
        raise TypeError(f"imwrite: unsupported sample type {a.dtype}")
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError(f"imwrite: unsupported array shape {np.asarray(img).shape}")
    H, W, S = a.shape
    if H == 0 or W == 0:
        raise ValueError("imwrite: empty image")
    a = np.ascontiguousarray(_swap_rb(a)).astype(a.dtype.newbyteorder("<"), copy=False)
    row_bytes = W * S * a.dtype.itemsize
    rps = max(1, min(H, (1 << 13) // row_bytes))
    n_strips = (H + rps - 1) // rps
    data_bytes = H * row_bytes
    big = data_bytes + 16 * n_strips + 4096 >= (1 << 32)
    osz = 8 if big else 4
    otype = 16 if big else 4
    header = 16 if big else 8
    counts = [row_bytes * min(rps, H - s * rps) for s in range(n_strips)]
    offsets = [header + s * rps * row_bytes for s in range(n_strips)]
    pos = header + data_bytes
    pos += pos & 1
    entries = []          # (tag, type, values)
    fmt = 3 if a.dtype.kind == "f" else 1
    entries.append((256, 4, [W]))
    entries.append((257, 4, [H]))
    entries.append((258, 3, [a.dtype.itemsize * 8] * S))
    entries.append((259, 3, [1]))
    entries.append((262, 3, [2 if S >= 3 else 1]))
    entries.append((273, otype, offsets))
    entries.append((277, 3, [S]))
    entries.append((278, 4, [rps]))
    entries.append((279, otype, counts))
    entries.append((284, 3, [1]))
    if S == 4:
        entries.append((338, 3, [2]))                   # unassociated alpha
    entries.append((339, 3, [fmt] * S))
    entries.sort(key=lambda e: e[0])
    n = len(entries)
    ifd_off = pos
    ifd_size = (8 + 20 * n + 8) if big else (2 + 12 * n + 4)
    extra_off = ifd_off + ifd_size
    ifd = bytearray()
    extra = bytearray()
    ifd += struct.pack("<Q" if big else "<H", n)
    for tag, typ, vals in entries:
        payload = struct.pack("<" + _TYPE_FMT[typ] * len(vals), *vals)
        ifd += struct.pack("<HH", tag, typ) + struct.pack("<Q" if big else "<I", len(vals))
        if len(payload) <= osz:
            ifd += payload.ljust(osz, b"\0")
        else:
            ifd += struct.pack("<Q" if big else "<I", extra_off + len(extra))
            extra += payload
            if len(extra) & 1:
                extra += b"\0"
    ifd += struct.pack("<Q" if big else "<I", 0)
    path = Path(path)
    with open(path, "wb") as f:
        if big:
            f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, ifd_off))
        else:
            f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(a.tobytes())
        if (header + data_bytes) & 1:
            f.write(b"\0")
        f.write(bytes(ifd))
        f.write(bytes(extra))
    return True
