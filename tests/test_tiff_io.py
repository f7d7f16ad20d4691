"""tiff_io on the CPU: the codec against Pillow/libtiff in both directions, the cv.imread / cv.imwrite
conventions the reference relies on, and malformed input. cv2 is absent and the reference ships no TIFF fixtures, so the
float64 three-channel files are checked by round trip only (parity unpinned against cv2, see tiff_io's docstring)."""
import struct

import numpy as np
import pytest

from camera_linearity_amd import tiff_io as T

Image = pytest.importorskip("PIL.Image")


def sample_image(h=123, w=211):
    rng = np.random.default_rng(0)
    base = (np.add.outer(np.arange(h), np.arange(w)) % 256).astype(np.uint8)
    img = np.stack([base, base[::-1], base // 3], -1)
    img[20:60, 20:70] = rng.integers(0, 256, (40, 50, 3))
    return img


@pytest.mark.parametrize("comp,info", [(None, None), ("tiff_lzw", None), ("tiff_lzw", {317: 2}), ("tiff_adobe_deflate", None),
                                       ("tiff_adobe_deflate", {317: 2}), ("packbits", None)])
def test_reads_what_libtiff_writes(tmp_path, comp, info):
    img = sample_image()
    p = tmp_path / "a.tif"
    kw = {}
    if comp:
        kw["compression"] = comp
    if info:
        kw["tiffinfo"] = info
    Image.fromarray(img).save(p, format="TIFF", **kw)           # file order RGB
    assert np.array_equal(T.read_tiff(p), img)
    bgr = T.imread(p)                                            # cv.imread: BGR
    assert bgr.dtype == np.uint8 and np.array_equal(bgr, img[:, :, ::-1])
    assert np.array_equal(T.imread(p, T.IMREAD_UNCHANGED), img[:, :, ::-1])


def test_multi_strip_lzw_large(tmp_path):
    """OpenCV-like layout: many ~8 KiB strips, LZW + horizontal predictor, decoded on the thread pool."""
    rng = np.random.default_rng(1)
    img = np.clip(np.cumsum(rng.integers(-3, 4, (700, 900, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    p = tmp_path / "b.tif"
    Image.fromarray(img).save(p, format="TIFF", compression="tiff_lzw", tiffinfo={317: 2, 278: 3})
    assert np.array_equal(T.read_tiff(p), img)


def test_gray_and_16bit(tmp_path):
    rng = np.random.default_rng(2)
    g8 = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    g16 = rng.integers(0, 65536, (64, 80)).astype(np.uint16)
    g16[:32] = np.arange(80) * 700
    p = tmp_path / "g.tif"
    Image.fromarray(g8).save(p, format="TIFF", compression="tiff_lzw")
    assert np.array_equal(T.imread(p, T.IMREAD_UNCHANGED), g8)
    c = T.imread(p)                                              # default flag: 3-channel
    assert c.shape == (40, 50, 3) and np.array_equal(c[:, :, 1], g8)
    Image.fromarray(g16).save(p, format="TIFF", compression="tiff_lzw", tiffinfo={317: 2})
    assert np.array_equal(T.imread(p, T.IMREAD_UNCHANGED), g16)
    assert np.array_equal(T.imread(p)[:, :, 0], (g16 >> 8).astype(np.uint8))


def test_libtiff_reads_what_we_write(tmp_path):
    img = sample_image()
    p = tmp_path / "w.tif"
    assert T.imwrite(p, img[:, :, ::-1])                         # BGR in, RGB in the file
    assert np.array_equal(np.array(Image.open(p)), img)
    g16 = (np.arange(64 * 80).reshape(64, 80) * 9 % 65536).astype(np.uint16)
    T.imwrite(p, g16)
    assert np.array_equal(np.array(Image.open(p)), g16)
    f32 = np.random.default_rng(3).random((33, 47)).astype(np.float32)
    T.imwrite(p, f32)
    assert np.array_equal(np.array(Image.open(p)), f32)
    rgba = np.random.default_rng(4).integers(0, 256, (9, 7, 4), dtype=np.uint8)
    T.imwrite(p, rgba[:, :, [2, 1, 0, 3]])
    assert np.array_equal(np.array(Image.open(p)), rgba)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.uint8, np.uint16])
@pytest.mark.parametrize("shape", [(33, 47, 3), (5, 4), (1, 1, 3), (300, 3000, 3)])
def test_round_trip(tmp_path, dtype, shape):
    """The reference's 8-bit / 64-bit round trip (tests/integration/test_integration_image_set.py:46-83) for the codec."""
    rng = np.random.default_rng(5)
    a = rng.random(shape) if np.dtype(dtype).kind == "f" else rng.integers(0, np.iinfo(dtype).max + 1, shape)
    a = a.astype(dtype)
    p = tmp_path / "r.tif"
    T.imwrite(p, a)
    b = T.imread(p, T.IMREAD_UNCHANGED)
    assert b.dtype == a.dtype and b.shape == a.shape and np.array_equal(a, b)


def test_big_endian_and_bigtiff(tmp_path):
    """Hand-built files: a big-endian classic TIFF and a little-endian BigTIFF of the same 2 x 3 RGB float64 image."""
    a = np.arange(18, dtype=np.float64).reshape(2, 3, 3) / 7
    p = tmp_path / "mm.tif"
    data = a.astype(">f8").tobytes()
    ents = [(256, 4, 1, 3), (257, 4, 1, 2), (258, 3, 3, None), (259, 3, 1, 1), (262, 3, 1, 2), (273, 4, 1, 8), (277, 3, 1, 3),
            (278, 4, 1, 2), (279, 4, 1, len(data)), (339, 3, 3, None)]
    ifd_off = 8 + len(data)
    extra_off = ifd_off + 2 + 12 * len(ents) + 4
    ifd = struct.pack(">H", len(ents))
    extra = b""
    for tag, typ, cnt, val in ents:
        if val is None:
            vals = [64] * 3 if tag == 258 else [3] * 3
            ifd += struct.pack(">HHII", tag, typ, cnt, extra_off + len(extra))
            extra += struct.pack(">HHH", *vals)
        elif typ == 3:
            ifd += struct.pack(">HHIHH", tag, typ, cnt, val, 0)
        else:
            ifd += struct.pack(">HHII", tag, typ, cnt, val)
    ifd += struct.pack(">I", 0)
    p.write_bytes(struct.pack(">2sHI", b"MM", 42, ifd_off) + data + ifd + extra)
    assert np.array_equal(T.read_tiff(p), a)
    assert np.array_equal(T.imread(p, T.IMREAD_UNCHANGED), a[:, :, ::-1])
    # BigTIFF
    data = a.astype("<f8").tobytes()
    ifd_off = 16 + len(data)
    n = len(ents)
    extra_off = ifd_off + 8 + 20 * n + 8
    ifd, extra = struct.pack("<Q", n), b""
    for tag, typ, cnt, val in ents:
        if tag == 273:
            val = 16
        if val is None:
            vals = [64] * 3 if tag == 258 else [3] * 3
            ifd += struct.pack("<HHQ", tag, typ, cnt) + struct.pack("<HHH", *vals).ljust(8, b"\0")      # 6 bytes fit inline
        elif typ == 3:
            ifd += struct.pack("<HHQ", tag, typ, cnt) + struct.pack("<H", val).ljust(8, b"\0")
        else:
            ifd += struct.pack("<HHQ", tag, typ, cnt) + struct.pack("<I", val).ljust(8, b"\0")
    ifd += struct.pack("<Q", 0)
    p.write_bytes(struct.pack("<2sHHHQ", b"II", 43, 8, 0, ifd_off) + data + ifd + extra)
    assert np.array_equal(T.read_tiff(p), a)


def test_missing_and_malformed(tmp_path):
    assert T.imread(tmp_path / "absent.tif") is None                      # cv.imread returns None
    p = tmp_path / "bad.tif"
    p.write_bytes(b"not a tiff at all")
    with pytest.raises(ValueError):
        T.imread(p)
    img = sample_image(80, 90)
    Image.fromarray(img).save(p, format="TIFF", compression="tiff_lzw")
    raw = bytearray(p.read_bytes())
    raw[8:400] = bytes(392)                                               # wipe the start of the LZW data
    p.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        T.read_tiff(p)
    with pytest.raises(TypeError):
        T.imwrite(p, np.zeros((4, 4), dtype=np.int64))
    with pytest.raises(ValueError):
        T.imwrite(p, np.zeros((4, 4, 2), dtype=np.uint8))


def test_lzw_decoder_direct():
    """hm_tiff_lzw_decode on a hand-assembled stream: Clear, 'A', 'B', 258 ('AB'), 260 (KwKwK: 'ABA'), EOI."""
    from camera_linearity_amd import _native as nat
    import ctypes as C
    codes = [256, 65, 66, 258, 260, 257]
    bits = "".join(format(c, "09b") for c in codes)
    bits += "0" * (-len(bits) % 8)
    src = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    out = (C.c_uint8 * 16)()
    n = nat.lib.hm_tiff_lzw_decode(src, len(src), out, 16)
    assert bytes(out[:n]) == b"ABABABA"
    assert nat.lib.hm_tiff_lzw_decode(src, len(src), out, 3) == nat.HM_ESHAPE
    pb = bytes([2, 1, 2, 3, 0xFE, 9, 0x80, 0, 7])                         # literal 1 2 3, 9 x3, no-op, literal 7
    n = nat.lib.hm_tiff_packbits_decode(pb, len(pb), out, 16)
    assert bytes(out[:n]) == bytes([1, 2, 3, 9, 9, 9, 7])


def test_truncated_and_garbage_directories(tmp_path):
    """Truncated files and directory entries that point outside the file are reported as ValueError, not crashes."""
    img = sample_image(60, 70)
    p = tmp_path / "t.tif"
    T.imwrite(p, img)
    raw = p.read_bytes()
    for cut in (6, 9, len(raw) // 2, len(raw) - 7):
        p.write_bytes(raw[:cut])
        with pytest.raises(ValueError):
            T.read_tiff(p)
    bad = bytearray(raw)
    ifd = struct.unpack_from("<I", raw, 4)[0]
    struct.pack_into("<I", bad, ifd + 2 + 4, 0x7FFFFFF0)            # first entry: absurd value count
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        T.read_tiff(p)
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 4, len(raw) + 1000)                   # IFD offset beyond the file
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        T.read_tiff(p)


def test_decoders_and_reader_survive_random_corruption(tmp_path):
    """The strip decoders (C, host side of libhdrmerge.so) on 10 000 random byte strings with random output capacities, and imread on 3 000
    randomly corrupted or truncated copies of an 8-bit and a float64 TIFF: a decoder never writes past its capacity, a damaged file is either
    read or refused with TiffError / NotImplementedError - nothing crashes, no other exception type escapes."""
    import ctypes as C
    import random
    from camera_linearity_amd import _native as nat
    lib = nat.hip_lib
    rng = random.Random(5)
    for _ in range(10000):
        n = rng.choice([0, 1, 2, 3, 8, 64, 300, 5000])
        src = bytes(rng.getrandbits(8) for _ in range(n)) if rng.random() < 0.7 else bytes([rng.choice([0x80, 0x00, 0xff, 0x01])] * n)
        cap = rng.choice([0, 1, 7, 64, 1000, 70000])
        dst = (C.c_uint8 * (max(cap, 1) + 16))()
        guard = bytes(dst[max(cap, 1):])
        for fn in (lib.hm_tiff_lzw_decode, lib.hm_tiff_packbits_decode):
            assert fn(src, len(src), dst, cap) <= cap
        assert bytes(dst[max(cap, 1):]) == guard                       # nothing beyond the capacity was touched
    img = np.random.default_rng(0).integers(0, 256, (37, 53, 3)).astype(np.uint8)
    f64 = np.random.default_rng(1).random((20, 31, 3))
    T.imwrite(tmp_path / "a.tif", img)
    T.imwrite(tmp_path / "b.tif", f64)
    outcomes = set()
    for name in ("a.tif", "b.tif"):
        raw = bytearray((tmp_path / name).read_bytes())
        for _ in range(1500):
            b = bytearray(raw)
            for _k in range(rng.choice([1, 1, 2, 5, 20])):
                b[rng.randrange(0, min(len(b), 400) if rng.random() < 0.7 else len(b))] = rng.getrandbits(8)
            if rng.random() < 0.2:
                b = b[:rng.randrange(0, len(b))]
            (tmp_path / "c.tif").write_bytes(bytes(b))
            try:
                T.imread(tmp_path / "c.tif", T.IMREAD_UNCHANGED)
                outcomes.add("read")
            except (T.TiffError, NotImplementedError) as e:
                outcomes.add(type(e).__name__)
    assert {"read", "TiffError"} <= outcomes
