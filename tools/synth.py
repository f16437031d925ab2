"""Synthetic inputs of the adaptive-warp / flow-projection operators.

numpy generators (seeded, CPU -- used by the parity tests so that the oracle and the HIP path see the very
same bytes) and torch generators (seeded, on-device -- used by bench.py at the full benchmark sizes, where a
host-side 2.8 GB batch would only measure PCIe).

  image   x    ~ U[0,1)
  flow    smooth: bilinear x16 upsample of N(0, 4^2) low-resolution noise (|f| mostly < 12 px -- what a flow
               field looks like after FlowProjection); iid: N(0, sigma^2) per pixel (defeats any tiling);
               video (device generator only): as smooth with a x64 upsample, i.e. four times smoother
  filter  k    ~ U[0,1) / fs^2   (taps sum to about 1/2: outputs stay O(1))
  depth   d    ~ U[0.1, 1.1)
  grad    g    ~ U[0,1)
"""
import numpy as np


def np_image(rng, B, C, H, W):
    return rng.random((B, C, H, W), dtype=np.float32)


def np_filter(rng, B, H, W, fs=4, scale=None):
    k = rng.random((B, fs * fs, H, W), dtype=np.float32)
    return k if scale is None else (k * np.float32(scale)).astype(np.float32)


def np_depth(rng, B, H, W):
    return (rng.random((B, 1, H, W), dtype=np.float32) + np.float32(0.1)).astype(np.float32)


def _upsample_bilinear(lo, H, W):
    """[B,2,h,w] -> [B,2,H,W], align_corners=True style bilinear resampling (pure numpy)."""
    B, C, h, w = lo.shape
    ys = np.linspace(0, h - 1, H, dtype=np.float64)
    xs = np.linspace(0, w - 1, W, dtype=np.float64)
    y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, h - 1); wy = (ys - y0)[None, None, :, None]
    x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, w - 1); wx = (xs - x0)[None, None, None, :]
    a = lo[:, :, y0][:, :, :, x0] * (1 - wx) + lo[:, :, y0][:, :, :, x1] * wx
    b = lo[:, :, y1][:, :, :, x0] * (1 - wx) + lo[:, :, y1][:, :, :, x1] * wx
    return (a * (1 - wy) + b * wy).astype(np.float32)


def np_flow(rng, B, H, W, kind="smooth", sigma=None):
    if kind == "smooth":
        s = 4.0 if sigma is None else sigma
        h, w = max(2, (H + 15) // 16 + 1), max(2, (W + 15) // 16 + 1)
        lo = rng.normal(0.0, s, (B, 2, h, w))
        return _upsample_bilinear(lo, H, W)
    if kind == "iid":
        s = 3.0 if sigma is None else sigma
        return rng.normal(0.0, s, (B, 2, H, W)).astype(np.float32)
    if kind == "zero":
        return np.zeros((B, 2, H, W), np.float32)
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------- torch (device)
def torch_inputs(device, B, C, H, W, fs=4, flow_kind="smooth", seed=1234, with_grad=False, with_depth=False):
    """Device-side generation for bench.py: dict with x, flow, filt (+ gout, depth)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    out["x"] = torch.rand((B, C, H, W), device=device, generator=g, dtype=torch.float32)
    if flow_kind == "smooth":
        h, w = (H + 15) // 16 + 1, (W + 15) // 16 + 1
        lo = torch.randn((B, 2, h, w), device=device, generator=g, dtype=torch.float32) * 4.0
        out["flow"] = F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=True).contiguous()
    elif flow_kind == "video":
        # same magnitude as "smooth", four times smoother (x64 upsample: ~0.09 px of flow change per pixel instead
        # of ~0.35) -- closer to what a flow network produces inside moving objects; secondary bench rows only
        h, w = (H + 63) // 64 + 1, (W + 63) // 64 + 1
        lo = torch.randn((B, 2, h, w), device=device, generator=g, dtype=torch.float32) * 4.0
        out["flow"] = F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=True).contiguous()
    elif flow_kind == "iid":
        out["flow"] = torch.randn((B, 2, H, W), device=device, generator=g, dtype=torch.float32) * 3.0
    else:
        raise ValueError(flow_kind)
    out["filt"] = torch.rand((B, fs * fs, H, W), device=device, generator=g, dtype=torch.float32) / (fs * fs)
    if with_grad:
        out["gout"] = torch.rand((B, C, H, W), device=device, generator=g, dtype=torch.float32)
    if with_depth:
        out["depth"] = torch.rand((B, 1, H, W), device=device, generator=g, dtype=torch.float32) + 0.1
    return out


def padded_planes(src, row_pad=64, plane_pad=0):
    """The same values as `src` ([B, C, H, W], fp32) in a layout whose rows are `row_pad` floats longer and whose channel planes are
    `plane_pad` floats further apart -- a VIEW the library takes as it is (unit w-stride; h / c / b strides are honoured as in
    my_lib.c:950-954).  Why: a 1280-wide fp32 row is 5 x 1024 B and a 720p plane 900 x 4096 B, so the sixteen rows of a tile and
    the sixteen tap planes of a site fall on FOUR of the sixteen 256-byte slots of a 4 KiB period; 64 more floats per row (or per
    plane) spread them over all sixteen and the RGB forward runs 7-9 % faster (profiles/r06_plane_strides.txt).  The reference's
    callers hand over contiguous tensors; a caller that owns its allocation can do this."""
    import torch
    b, c, h, w = src.shape
    buf = torch.empty((b, c, h * (w + row_pad) + plane_pad), device=src.device, dtype=src.dtype)
    view = buf[:, :, :h * (w + row_pad)].view(b, c, h, w + row_pad)[:, :, :, :w]
    view.copy_(src)
    return view
