"""Frame-pair inference the way the reference demos run it (the part with a bearing on the
hot path): replicate-pad both frames so that H and W are multiples of 128 -- or by 32 on each side when they
already are -- run the network, crop the padding off (demo_HD720p.py:88-113,138-146; demo_MiddleBury.py:74-110).
The HD demo's YUV 4:2:0 reader / writer and its per-pair loop are in yuv_io.py (numpy only); the still-image demo's
files are read and written by png_io.py (zlib + numpy): 8-bit grey, grey + alpha, RGB, RGBA and palette PNGs, all five
scanline filters, CRC-checked -- no tRNS chunk (palette transparency is ignored), no 16-bit samples, no interlacing
(both rejected with an error).
"""
import torch
import torch.nn.functional as F


def pad_amounts(height, width):
    """(left, right, top, bottom) of the reference rule, demo_HD720p.py:88-106."""
    def one(n):
        if n != ((n >> 7) << 7):
            total = (((n >> 7) + 1) << 7) - n
            first = int(total / 2)
            return first, total - first
        return 32, 32
    left, right = one(width)
    top, bottom = one(height)
    return left, right, top, bottom


def interpolate_pairs(model, frame0, frame2, which=1):
    """frame0, frame2: [B, 3, H, W] in [0, 1].  Returns the interpolated middle frames [B, 3, H, W]
    (which = 1: the rectified output, 0: the blended one -- the demos' `save_which`), padding removed."""
    assert frame0.shape == frame2.shape and frame0.dim() == 4
    h, w = frame0.shape[2], frame0.shape[3]
    left, right, top, bottom = pad_amounts(h, w)
    x = torch.stack([F.pad(f, (left, right, top, bottom), mode="replicate") for f in (frame0, frame2)])
    with torch.no_grad():
        frames, _flows, _filters, _occlusions = model(x)
    return frames[which][:, :, top:top + h, left:left + w]
