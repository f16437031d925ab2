"""Planar YUV 4:2:0 frame files and the demo loop around a frame pair.

What the reference's HD demo does around the network (yuv_frame_io.py:31-200, demo_HD720p.py:60-170), restated on
numpy alone -- the reference needs scipy.misc.imresize and skimage.color for it, neither of which exists here:

  * a frame of an I420 file is h*w bytes of Y, then (h/2)*(w/2) of U, then of V (yuv_frame_io.py:38-41,50-52);
  * reading: chroma is upsampled by pixel repetition (imresize(..., interp='nearest') of an exact factor two picks
    source index i // 2), Y/255, U/255 - 0.5, V/255 - 0.5 go through the inverse of the analog-YUV matrix that
    skimage calls yuv2rgb, clipped to [0, 1], times 255, truncated to uint8 (:65-66,85-89);
  * writing: RGB/255 through the forward matrix, U + 0.5 and V + 0.5 clipped to [0, 1], every second chroma sample of
    every second row kept, times 255, truncated to uint8 (:129-148);
  * the demo reads frames i and i + 2, interpolates the middle one (pad to multiples of 128, crop: inference.py),
    writes frame i and the rounded result, and scores the result against the real frame i + 1 on the truncated luma
    planes: mean absolute error and PSNR (demo_HD720p.py:75-167).
"""
import numpy as np

# skimage.color's yuv_from_rgb (the analog YUV of PAL: Y = 0.299 R + 0.587 G + 0.114 B, U = 0.492 (B - Y), V = 0.877 (R - Y))
YUV_FROM_RGB = np.array([[0.299, 0.587, 0.114],
                         [-0.14714119, -0.28886916, 0.43601035],
                         [0.61497538, -0.51496512, -0.10001026]], dtype=np.float64)
RGB_FROM_YUV = np.linalg.inv(YUV_FROM_RGB)


def rgb_to_yuv(rgb):
    """[..., 3] float RGB in [0, 1] -> analog YUV (what skimage.color.rgb2yuv computes)."""
    return np.asarray(rgb, dtype=np.float64) @ YUV_FROM_RGB.T


def yuv_to_rgb(yuv):
    return np.asarray(yuv, dtype=np.float64) @ RGB_FROM_YUV.T


def frame_bytes(h, w):
    return h * w + 2 * ((h // 2) * (w // 2))


class Yuv420Reader(object):
    """reader.read(k) -> (frame, True) or (None, False) past the end; frame is uint8 RGB [h, w, 3] (to_rgb) or the
    uint8 Y, U, V planes stacked at full resolution."""

    def __init__(self, path, h, w, to_rgb=True):
        if h % 2 or w % 2:
            raise ValueError("4:2:0 needs even dimensions, got %dx%d" % (w, h))
        self.h, self.w, self.to_rgb = h, w, to_rgb
        self.frame_length = frame_bytes(h, w)
        self.fp = open(path, "rb")

    def read(self, frame_index=None):
        h, w = self.h, self.w
        if frame_index is not None:
            self.fp.seek(frame_index * self.frame_length, 0)
        buf = self.fp.read(self.frame_length)
        if len(buf) < self.frame_length:
            return None, False
        a = np.frombuffer(buf, dtype=np.uint8)
        ny, nc = h * w, (h // 2) * (w // 2)
        y = a[:ny].reshape(h, w)
        u = a[ny:ny + nc].reshape(h // 2, w // 2).repeat(2, axis=0).repeat(2, axis=1)
        v = a[ny + nc:].reshape(h // 2, w // 2).repeat(2, axis=0).repeat(2, axis=1)
        if not self.to_rgb:
            return np.stack((y, u, v), axis=-1), True
        yuv = np.stack((y / 255.0, u / 255.0 - 0.5, v / 255.0 - 0.5), axis=-1)
        return (255.0 * np.clip(yuv_to_rgb(yuv), 0.0, 1.0)).astype(np.uint8), True

    def close(self):
        self.fp.close()


class Yuv420Writer(object):
    """writer.write(frame): uint8 RGB [h, w, 3] (from_rgb) or full-resolution Y, U, V planes stacked."""

    def __init__(self, path, from_rgb=True):
        self.fp = open(path, "wb")
        self.from_rgb = from_rgb

    def write(self, frame):
        frame = np.asarray(frame)
        if frame.ndim != 3 or frame.shape[2] != 3 or frame.shape[0] % 2 or frame.shape[1] % 2:
            raise ValueError("expected [h, w, 3] with even h and w, got %s" % (frame.shape,))
        if self.from_rgb:
            yuv = rgb_to_yuv(frame / 255.0)
            y = (255.0 * yuv[:, :, 0]).astype(np.uint8)
            u = (255.0 * np.clip(yuv[::2, ::2, 1] + 0.5, 0.0, 1.0)).astype(np.uint8)
            v = (255.0 * np.clip(yuv[::2, ::2, 2] + 0.5, 0.0, 1.0)).astype(np.uint8)
        else:
            y, u, v = frame[:, :, 0], frame[::2, ::2, 1], frame[::2, ::2, 2]
        for plane in (y, u, v):
            self.fp.write(np.ascontiguousarray(plane).tobytes())
        return True

    def close(self):
        self.fp.close()


def luma_scores(rec_rgb, gt_rgb):
    """The demo's two numbers for one interpolated frame: mean |dY| and PSNR on the truncated 8-bit luma planes
    (demo_HD720p.py:150-167; PSNR 100 for identical planes)."""
    gy = (rgb_to_yuv(gt_rgb / 255.0)[:, :, 0] * 255.0).astype(np.uint8).astype(np.float64)
    ry = (rgb_to_yuv(rec_rgb / 255.0)[:, :, 0] * 255.0).astype(np.uint8).astype(np.float64)
    diff = ry - gy
    mse = float(np.mean(diff ** 2))
    psnr = 100.0 if mse == 0 else float(20.0 * np.log10(255.0 / np.sqrt(mse)))
    return float(np.mean(np.abs(diff))), psnr


def interpolate_yuv_sequence(model, in_path, out_path, h, w, device, first=0, last=100, pairs_per_step=1, which=1):
    """The demo loop: for i = first, first + 2, ...: frames i and i + 2 in, the interpolated frame i + 1 out
    (out_path receives frame i, then the interpolated frame), scored against the file's own frame i + 1.
    pairs_per_step > 1 batches that many independent pairs through the network at once (same results: every operator
    of the path is per frame pair).  Returns the list of (index of the interpolated frame, mean |dY|, PSNR)."""
    import torch
    from .inference import interpolate_pairs
    reader, writer = Yuv420Reader(in_path, h, w, to_rgb=True), Yuv420Writer(out_path, from_rgb=True)
    scores = []
    try:
        index = first
        while index < last:
            batch = []
            while len(batch) < pairs_per_step and index < last:
                a, ok_a = reader.read(index)
                b, ok_b = reader.read(index + 2)
                if not (ok_a and ok_b):
                    index = last
                    break
                batch.append((index, a, b))
                index += 2
            if not batch:
                break

            def to_tensor(frames):
                x = np.stack([np.transpose(f, (2, 0, 1)) for f in frames]).astype(np.float32) / 255.0
                return torch.from_numpy(x).to(device)
            mid = interpolate_pairs(model, to_tensor([a for _, a, _ in batch]), to_tensor([b for _, _, b in batch]), which)
            mid = np.round(255.0 * mid.clamp(0.0, 1.0).cpu().numpy()).astype(np.uint8)
            for k, (i, a, _b) in enumerate(batch):
                rec = np.transpose(mid[k], (1, 2, 0))
                writer.write(a)
                writer.write(rec)
                gt, ok = reader.read(i + 1)
                if ok:
                    scores.append((i + 1,) + luma_scores(rec, gt))
    finally:
        reader.close()
        writer.close()
    return scores
