"""8-bit PNG files and the demo loop over a Middlebury-style tree.

The reference's still-image demo (demo_MiddleBury.py:66-181) reads `<data>/<scene>/frame10.png` and `frame11.png` with
scipy.misc.imread, interpolates the middle frame (pad to multiples of 128, crop: inference.py), writes it with
scipy.misc.imsave and scores it against `<gt>/<scene>/frame10i11.png`: mean absolute RGB error and PSNR on the 8-bit
images, plus a difference picture 128 + rec - gt cast to uint8.  Neither scipy.misc nor an image library exists here,
so the codec is restated on zlib + numpy: the PNG subset those files use -- 8 bits per sample, grey / grey+alpha /
RGB / RGBA / palette, non-interlaced, all five scanline filters on reading; the writer emits RGB or grey with the
"up" filter, which such pictures compress well under.
"""
import os
import struct
import zlib

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}          # colour type -> samples per pixel


def _chunks(buf):
    pos = len(_SIGNATURE)
    while pos + 8 <= len(buf):
        n, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + n]
        if len(body) < n:
            raise ValueError("truncated PNG chunk %r" % kind)
        (crc,) = struct.unpack(">I", buf[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(kind + body) & 0xFFFFFFFF != crc:
            raise ValueError("bad CRC in PNG chunk %r" % kind)
        yield kind, body
        pos += 12 + n


def _unfilter(raw, h, w, bpp):
    """Undo the per-scanline filters (PNG specification, section 9): raw is h rows of 1 + w * bpp bytes."""
    stride = w * bpp
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, stride + 1)
    out = np.zeros((h + 1, stride + bpp), dtype=np.uint8)     # a zero row above, bpp zero bytes to the left
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:]
        cur, up = out[y + 1], out[y]
        if ft == 0:
            cur[bpp:] = line
        elif ft == 2:
            cur[bpp:] = line + up[bpp:]
        elif ft == 1:                                         # left neighbour: a running sum per byte lane, mod 256
            lanes = line.reshape(w, bpp).astype(np.uint32)
            cur[bpp:] = (np.cumsum(lanes, axis=0) & 0xFF).astype(np.uint8).reshape(-1)
        elif ft in (3, 4):                                    # average / Paeth depend on the byte just produced:
            c, u, ln = [0] * (stride + bpp), up.tolist(), line.tolist()     # a plain loop, on Python ints
            if ft == 3:
                for i in range(stride):
                    c[i + bpp] = (ln[i] + ((c[i] + u[i + bpp]) >> 1)) & 0xFF
            else:
                for i in range(stride):
                    a, b, d = c[i], u[i + bpp], u[i]          # left, up, upper left (u carries bpp bytes of padding)
                    p = a + b - d
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - d)
                    c[i + bpp] = (ln[i] + (a if (pa <= pb and pa <= pc) else (b if pb <= pc else d))) & 0xFF
            cur[:] = np.array(c, dtype=np.uint8)
        else:
            raise ValueError("unknown PNG filter type %d" % ft)
    return out[1:, bpp:]


def read_png(path):
    """-> uint8 array [h, w] (grey) or [h, w, c] (c = 2, 3, 4); palette images come back as RGB."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != _SIGNATURE:
        raise ValueError("%s is not a PNG file" % path)
    header, palette, data = None, None, []
    for kind, body in _chunks(buf):
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            data.append(body)
        elif kind == b"IEND":
            break
    if header is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _comp, _filt, interlace = header
    if depth != 8 or ctype not in _CHANNELS or interlace != 0:
        raise ValueError("unsupported PNG: bit depth %d, colour type %d, interlace %d" % (depth, ctype, interlace))
    bpp = _CHANNELS[ctype]
    raw = zlib.decompress(b"".join(data))
    if len(raw) != h * (w * bpp + 1):
        raise ValueError("PNG data size does not match its header")
    img = _unfilter(raw, h, w, bpp).reshape(h, w, bpp)
    if ctype == 3:
        if palette is None:
            raise ValueError("palette PNG without PLTE")
        return palette[img[:, :, 0]]
    return img[:, :, 0].copy() if bpp == 1 else img.copy()


def write_png(path, image):
    """image: uint8 [h, w] or [h, w, 1 | 3 | 4]."""
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise ValueError("write_png takes uint8, got %s" % a.dtype)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError("expected [h, w] or [h, w, 1 | 3 | 4], got %s" % (a.shape,))
    h, w, c = a.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    flat = a.reshape(h, w * c)
    up = np.zeros_like(flat)
    up[1:] = flat[:-1]
    rows = np.empty((h, w * c + 1), dtype=np.uint8)
    rows[:, 0] = 2                                            # filter type "up"
    rows[:, 1:] = flat - up                                   # uint8 arithmetic wraps modulo 256, as the filter asks

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(_SIGNATURE)
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def rgb_scores(rec_rgb, gt_rgb):
    """The still-image demo's numbers (demo_MiddleBury.py:164-172): mean |rec - gt| over all samples and PSNR; also the
    difference picture it saves (128 + rec - gt, cast to uint8 the way numpy casts: modulo 256)."""
    diff = 128.0 + np.asarray(rec_rgb, dtype=np.float64) - np.asarray(gt_rgb, dtype=np.float64)
    err = float(np.mean(np.abs(diff - 128.0)))
    mse = float(np.mean((diff - 128.0) ** 2))
    psnr = 100.0 if mse == 0 else min(100.0, float(20.0 * np.log10(255.0 / np.sqrt(mse))))    # (capped like the luma scores:
                                                                                              #  an exact match must not make the scene average inf)
    return err, psnr, (diff.astype(np.int64) & 0xFF).astype(np.uint8)


def interpolate_png_tree(model, data_dir, out_dir, device, gt_dir=None, first="frame10.png", second="frame11.png",
                         middle="frame10i11.png", which=1):
    """For every scene directory of data_dir: first + second in, the interpolated middle frame out (out_dir/<scene>/),
    scored against gt_dir/<scene>/<middle> where that exists.  Returns [(scene, mean abs error, PSNR)] (None, None
    without ground truth).  Scenes that are not three-channel are skipped, as the reference skips them."""
    import torch
    from .inference import interpolate_pairs
    results = []
    for scene in sorted(os.listdir(data_dir)):
        a_path, b_path = os.path.join(data_dir, scene, first), os.path.join(data_dir, scene, second)
        if not (os.path.isfile(a_path) and os.path.isfile(b_path)):
            continue
        a, b = read_png(a_path), read_png(b_path)
        if a.shape != b.shape:
            raise ValueError("%s: the two frames differ in size (%s, %s)" % (scene, a.shape, b.shape))
        if a.ndim != 3 or a.shape[2] != 3:
            continue

        def to_tensor(img):
            return torch.from_numpy(np.transpose(img, (2, 0, 1)).astype(np.float32) / 255.0).unsqueeze(0).to(device)
        mid = interpolate_pairs(model, to_tensor(a), to_tensor(b), which)
        rec = np.round(255.0 * mid.clamp(0.0, 1.0)[0].cpu().numpy()).astype(np.uint8)
        rec = np.ascontiguousarray(np.transpose(rec, (1, 2, 0)))
        os.makedirs(os.path.join(out_dir, scene), exist_ok=True)
        write_png(os.path.join(out_dir, scene, middle), rec)
        gt_path = os.path.join(gt_dir, scene, middle) if gt_dir else None
        if gt_path and os.path.isfile(gt_path):
            gt = read_png(gt_path)
            if gt.ndim == 3 and gt.shape[2] == 4:
                gt = gt[:, :, :3]
            err, psnr, diff = rgb_scores(rec, gt)
            stem = middle[:-4] if middle.lower().endswith(".png") else middle
            write_png(os.path.join(out_dir, scene, "%s_diff%.4f.png" % (stem, err)), diff)
            results.append((scene, err, psnr))
        else:
            results.append((scene, None, None))
    return results
