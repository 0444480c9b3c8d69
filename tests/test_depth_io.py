"""include/sobfu_amd/depth_io.hpp on the CPU: 16-bit PNG (all five row filters, split IDAT), PGM, raw readers and the .npy writer."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from sobfu_amd import build_host


def png_bytes(img, filters, bit_depth=16, idat_split=3):
    """Minimal PNG encoder with a chosen filter type per row (the decoder under test must undo every one of them)."""
    h, w = img.shape
    bpp = bit_depth // 8
    rows = img.astype(">u2").view(np.uint8).reshape(h, w * 2) if bpp == 2 else img.astype(np.uint8).reshape(h, w)
    rows = rows.astype(np.int32)
    prev = np.zeros(w * bpp, np.int32)
    raw = bytearray()
    for y in range(h):
        cur, ft = rows[y], filters[y % len(filters)]
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - a
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            out = cur - pred
        raw.append(ft)
        raw += bytes((out & 255).astype(np.uint8))
        prev = cur
    comp = zlib.compress(bytes(raw), 6)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, 0, 0, 0, 0)) + chunk(b"tEXt", b"k\x00v")
    step = max(1, len(comp) // idat_split)
    for i in range(0, len(comp), step):
        out += chunk(b"IDAT", comp[i:i + step])
    return out + chunk(b"IEND", b"")


@pytest.fixture(scope="module")
def tool():
    return build_host.build_io_tool()


def decode(tool, path, rows, cols, tmp_path):
    out = str(tmp_path / "out.raw")
    r = subprocess.run([tool, "read", str(path), str(rows), str(cols), out], capture_output=True, text=True)
    return r.returncode, (np.fromfile(out, np.uint16).reshape(rows, cols) if r.returncode == 0 else r.stdout)


def test_png_pgm_raw_roundtrip(tool, tmp_path):
    rng = np.random.default_rng(3)
    rows, cols = 37, 53
    img = rng.integers(0, 65536, (rows, cols)).astype(np.uint16)
    img[5:20, 7:30] = (np.arange(23)[None, :] * 40 + 900).astype(np.uint16)  # smooth patch: exercises the predictors
    for filters in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
        p = tmp_path / "d.png"
        p.write_bytes(png_bytes(img, filters))
        rc, got = decode(tool, p, rows, cols, tmp_path)
        assert rc == 0 and np.array_equal(got, img), filters
    p = tmp_path / "d8.png"
    p.write_bytes(png_bytes(img & 255, [4, 3], bit_depth=8))
    rc, got = decode(tool, p, rows, cols, tmp_path)
    assert rc == 0 and np.array_equal(got, img & 255)
    p = tmp_path / "d.pgm"
    p.write_bytes(b"P5\n%d %d\n65535\n" % (cols, rows) + img.astype(">u2").tobytes())
    rc, got = decode(tool, p, rows, cols, tmp_path)
    assert rc == 0 and np.array_equal(got, img)
    p = tmp_path / "d.raw"
    img.tofile(p)
    rc, got = decode(tool, p, rows, cols, tmp_path)
    assert rc == 0 and np.array_equal(got, img)


def test_reader_rejects_bad_input(tool, tmp_path):
    img = np.arange(12, dtype=np.uint16).reshape(3, 4)
    good = png_bytes(img, [1])
    p = tmp_path / "bad.png"
    p.write_bytes(good[:40] + bytes([good[40] ^ 1]) + good[41:])  # flips a bit inside IHDR/first chunks: CRC must catch it
    rc, msg = decode(tool, p, 3, 4, tmp_path)
    assert rc != 0 and "error" in msg
    p.write_bytes(good)
    rc, msg = decode(tool, p, 4, 3, tmp_path)  # wrong geometry
    assert rc != 0
    (tmp_path / "short.raw").write_bytes(b"\x00" * 10)
    rc, msg = decode(tool, tmp_path / "short.raw", 3, 4, tmp_path)
    assert rc != 0
    rc, msg = decode(tool, tmp_path / "missing.png", 3, 4, tmp_path)
    assert rc != 0


def test_npy_writer(tool, tmp_path):
    for shape in ((5,), (3, 4), (2, 3, 4, 4)):
        out = tmp_path / "a.npy"
        assert subprocess.run([tool, "npy", str(out), *map(str, shape)]).returncode == 0
        a = np.load(out)
        assert a.dtype == np.float32 and a.shape == shape
        assert np.array_equal(a.ravel(), 0.5 * np.arange(a.size, dtype=np.float32))


def test_pgm_comments_and_malformed_headers(tool, tmp_path):
    """PGM headers may carry '#' comment lines; short / malformed headers must be rejected without reading past the buffer"""
    rows, cols = 5, 7
    img = (np.arange(rows * cols, dtype=np.uint16).reshape(rows, cols) * 911 + 300).astype(np.uint16)
    body = img.astype(">u2").tobytes()
    p = tmp_path / "c.pgm"
    p.write_bytes(b"P5\n# made by a depth camera\n%d %d\n# maxval next\n65535\n" % (cols, rows) + body)
    rc, got = decode(tool, p, rows, cols, tmp_path)
    assert rc == 0 and np.array_equal(got, img)
    p.write_bytes(b"P5 %d\t%d 65535 " % (cols, rows) + body)  # any single whitespace byte may end the header
    rc, got = decode(tool, p, rows, cols, tmp_path)
    assert rc == 0 and np.array_equal(got, img)
    for bad in (b"P5", b"P5\n", b"P5\n7", b"P5\n7 5", b"P5\n7 5\n65535", b"P5\n7 x\n65535\n", b"P5\n# only a comment",
                b"P5\n99999999999 5\n65535\n", b"P5\n7 5\n255\n" + body, b"P5\n7 5\n65535\n" + body[:-1]):
        p.write_bytes(bad)
        rc, msg = decode(tool, p, rows, cols, tmp_path)
        assert rc != 0 and "error" in msg, bad


def test_ini_reader_requires_the_keys_the_reference_requires(tmp_path):
    """TSDF_TRUNC_DIST / ETA / VOL_POSE_T_Z are read unconditionally by the reference (src/apps/demo.cpp:71-74): a file without
    them, or with non-positive dims / size / truncation, is refused before anything touches the GPU"""
    app = build_host.build_app()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    good = open(os.path.join(root, "params", "config1_sphere_64.ini")).read()

    def run(text):
        p = tmp_path / "p.ini"
        p.write_text(text)
        return subprocess.run([app, str(p), "--synthetic", "1"], capture_output=True, text=True)

    for key in ("TSDF_TRUNC_DIST", "ETA", "VOL_POSE_T_Z"):
        r = run("\n".join(ln for ln in good.splitlines() if not ln.startswith(key + "=")))
        assert r.returncode == 2 and key in r.stdout, (key, r.stdout)
    for key, val in (("VOL_DIMS_Y", "0"), ("VOL_SIZE_Z", "-1"), ("TSDF_TRUNC_DIST", "0")):
        r = run("\n".join(f"{key}={val}" if ln.startswith(key + "=") else ln for ln in good.splitlines()))
        assert r.returncode == 2 and "positive" in r.stdout, (key, r.stdout)
    r = subprocess.run([app, str(tmp_path / "missing.ini"), "--synthetic", "1"], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stdout
