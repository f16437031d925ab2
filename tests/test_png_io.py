"""The still-image demo's PNG files and its loop over a scene tree (networks/png_io.py; reference demo_MiddleBury.py:66-181),
on the CPU: the codec against files filtered row by row with the specification's five filters (written out here, as an
encoder, independently of the decoder), a known file, round trips, the scores."""
import base64
import importlib.util
import os
import struct
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "memc-net_amd")


def _load(name):
    # networks/__init__ imports the HIP-backed operators; the IO helpers are plain numpy: load them by path
    if "networks_cpu" not in sys.modules:
        pkg = importlib.util.module_from_spec(importlib.util.spec_from_loader("networks_cpu", loader=None, is_package=True))
        pkg.__path__ = [os.path.join(PKG, "networks")]
        sys.modules["networks_cpu"] = pkg
    full = "networks_cpu." + name
    if full not in sys.modules:
        spec = importlib.util.spec_from_file_location(full, os.path.join(PKG, "networks", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    return sys.modules[full]


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _encode(img, ctype, row_filters, idat_pieces=1, palette=None):
    """A PNG encoder by the book (specification section 9.2): row y is filtered with row_filters[y % len]."""
    h, w, bpp = img.shape
    flat = img.reshape(h, w * bpp).astype(np.int64)
    raw = bytearray()
    for y in range(h):
        ft = row_filters[y % len(row_filters)]
        raw.append(ft)
        for i in range(w * bpp):
            x = int(flat[y, i])
            a = int(flat[y, i - bpp]) if i >= bpp else 0
            b = int(flat[y - 1, i]) if y > 0 else 0
            c = int(flat[y - 1, i - bpp]) if (y > 0 and i >= bpp) else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            raw.append((x - pred) & 0xFF)
    z = zlib.compress(bytes(raw), 9)
    step = (len(z) + idat_pieces - 1) // idat_pieces
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
    out += _chunk(b"tEXt", b"Comment\x00made by the test")          # an ancillary chunk the reader must skip
    if palette is not None:
        out += _chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    for i in range(0, len(z), step):
        out += _chunk(b"IDAT", z[i:i + step])
    return out + _chunk(b"IEND", b"")


@pytest.mark.parametrize("ctype,bpp", [(0, 1), (2, 3), (4, 2), (6, 4)])
@pytest.mark.parametrize("filters", [(0,), (1,), (2,), (3,), (4,), (4, 1, 3, 2, 0)])
def test_reader_undoes_every_filter(tmp_path, ctype, bpp, filters):
    P = _load("png_io")
    rng = np.random.default_rng(ctype * 10 + len(filters) + filters[0])
    img = rng.integers(0, 256, size=(9, 13, bpp), dtype=np.uint8)
    img[2:5, 3:9] = img[2, 3]                                       # a flat patch: predictions that hit exactly
    path = tmp_path / "f.png"
    path.write_bytes(_encode(img, ctype, filters, idat_pieces=3))
    got = P.read_png(str(path))
    want = img[:, :, 0] if bpp == 1 else img
    assert got.dtype == np.uint8 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_palette_and_known_file(tmp_path):
    P = _load("png_io")
    pal = np.array([[0, 0, 0], [255, 0, 0], [0, 255, 0], [10, 20, 30]])
    idx = np.array([[0, 1, 2], [3, 3, 1]], dtype=np.uint8)[:, :, None]
    path = tmp_path / "p.png"
    path.write_bytes(_encode(idx, 3, (1, 4), palette=pal))
    assert np.array_equal(P.read_png(str(path)), pal[idx[:, :, 0]].astype(np.uint8))
    # a 1 x 1 RGBA file that circulates widely as a data URI: IDAT inflates to 01 ff 00 00 7f (filter "sub")
    dot = base64.b64decode("iVBORw0KGgoAAAANSUhEUgAAAAEAAAABCAYAAAAfFcSJAAAADUlEQVR42mP8z8BQDwAEhQGAhKmMIQAAAABJRU5ErkJggg==")
    path = tmp_path / "dot.png"
    path.write_bytes(dot)
    assert np.array_equal(P.read_png(str(path)), np.array([[[255, 0, 0, 127]]], dtype=np.uint8))


def test_round_trip_and_rejections(tmp_path):
    P = _load("png_io")
    rng = np.random.default_rng(3)
    for shape in ((7, 5), (6, 10, 3), (4, 4, 4), (1, 1, 3), (33, 64, 3)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        path = str(tmp_path / "r.png")
        P.write_png(path, img)
        assert np.array_equal(P.read_png(path), img)
    with pytest.raises(ValueError):
        P.write_png(str(tmp_path / "x.png"), np.zeros((4, 4, 3), dtype=np.float32))
    bad = bytearray(open(str(tmp_path / "r.png"), "rb").read())
    bad[40] ^= 0xFF                                                  # a flipped byte inside IDAT: the CRC must catch it
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        P.read_png(str(tmp_path / "bad.png"))
    (tmp_path / "no.png").write_bytes(b"not a png at all")
    with pytest.raises(ValueError):
        P.read_png(str(tmp_path / "no.png"))
    sixteen = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 16, 2, 0, 0, 0)) + _chunk(b"IEND", b"")
    (tmp_path / "s.png").write_bytes(sixteen)
    with pytest.raises(ValueError):
        P.read_png(str(tmp_path / "s.png"))


def test_scores():
    """demo_MiddleBury.py:164-172 on two small pictures."""
    P = _load("png_io")
    gt = np.full((2, 2, 3), 100, dtype=np.uint8)
    rec = gt.copy()
    err, psnr, diff = P.rgb_scores(rec, gt)
    assert err == 0.0 and psnr == 100.0 and np.all(diff == 128)          # capped: an exact match must not make an average inf
    rec[0, 0] = (110, 90, 100)
    err, psnr, diff = P.rgb_scores(rec, gt)
    assert err == pytest.approx(20.0 / 12.0)
    assert psnr == pytest.approx(20.0 * np.log10(255.0 / np.sqrt(200.0 / 12.0)))
    assert tuple(diff[0, 0]) == (138, 118, 128)
    far = np.zeros((1, 1, 3), dtype=np.uint8)
    _, _, diff = P.rgb_scores(far, np.full((1, 1, 3), 200, dtype=np.uint8))
    assert int(diff[0, 0, 0]) == (128 - 200) % 256                   # the cast wraps, like the reference's astype("uint8")
