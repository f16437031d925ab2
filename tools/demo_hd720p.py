#!/usr/bin/env python
"""tools/demo_hd720p.py -- the reference's HD demo (demo_HD720p.py) on this repository's operators: interpolate the odd
frames of a planar YUV 4:2:0 file from its even ones and score them against the file's own odd frames.

    python tools/demo_hd720p.py --input clip_1280x720.yuv --output clip_interp.yuv [--height 720 --width 1280]
                                [--model MEMC_Net_star --weights best.pth --align-corners] [--first 0 --last 100]
                                [--pairs-per-step 4]

Frames i and i + 2 go in (replicate-padded to multiples of 128 like demo_HD720p.py:88-113), the output file receives
frame i and the interpolated frame i + 1 (demo_HD720p.py:146-149); printed per frame: mean |dY| and PSNR on the 8-bit
luma planes (:150-167).  Without --weights the network runs on its random initialisation (plumbing check only).
Needs a GPU: the operators have no CPU path.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "memc-net_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--model", default="MEMC_Net_star", choices=["MEMC_Net_star", "MEMC_Net"])
    ap.add_argument("--weights", default="")
    ap.add_argument("--align-corners", action="store_true",
                    help="bilinear upsampling as PyTorch 0.2 did it: what the published checkpoints were trained with")
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--last", type=int, default=100)
    ap.add_argument("--pairs-per-step", type=int, default=1)
    ap.add_argument("--save-which", type=int, default=1, choices=[0, 1], help="0: blended, 1: rectified (the demos' save_which)")
    a = ap.parse_args(argv)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP operators have no CPU fallback")
    import networks
    dev = torch.device("cuda", 0)
    net = getattr(networks, a.model)(channel=3, filter_size=4, training=False, align_corners=a.align_corners)
    if a.weights:
        state = torch.load(a.weights, map_location="cpu")
        net.load_state_dict(state.get("state_dict", state), strict=True)
    net = net.to(dev).eval()
    scores = networks.interpolate_yuv_sequence(net, a.input, a.output, a.height, a.width, dev, first=a.first, last=a.last,
                                               pairs_per_step=a.pairs_per_step, which=a.save_which)
    for index, err, psnr in scores:
        print("frame %4d  mean|dY| %.4f  PSNR %.3f dB" % (index, err, psnr))
    if scores:
        print("average over %d interpolated frames: mean|dY| %.4f  PSNR %.3f dB" % (
            len(scores), sum(s[1] for s in scores) / len(scores), sum(s[2] for s in scores) / len(scores)))


if __name__ == "__main__":
    main()
