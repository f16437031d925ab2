#!/usr/bin/env python
"""tools/demo_middlebury.py -- the reference's still-image demo (demo_MiddleBury.py) on this repository's operators: for every
scene directory, interpolate the frame between frame10.png and frame11.png and score it against the ground truth.

    python tools/demo_middlebury.py --data other-data --gt other-gt-interp --output results
                                    [--model MEMC_Net_star --weights best.pth --align-corners] [--save-which 1]

Each pair is replicate-padded to multiples of 128 like demo_MiddleBury.py:98-117 and cropped back; results/<scene>/ receives
frame10i11.png and, where ground truth exists, the difference picture 128 + rec - gt (named after the scene's mean
absolute error, :179); printed per scene: mean absolute RGB error and PSNR on the 8-bit pictures (:164-172).  PNG files
are read and written by networks/png_io.py (8 bits per sample, non-interlaced).  Without --weights the network runs on
its random initialisation (plumbing check only).  Needs a GPU: the operators have no CPU path.
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
    ap.add_argument("--data", required=True, help="directory of scene directories holding frame10.png / frame11.png")
    ap.add_argument("--gt", default="", help="directory of scene directories holding frame10i11.png")
    ap.add_argument("--output", required=True)
    ap.add_argument("--model", default="MEMC_Net_star", choices=["MEMC_Net_star", "MEMC_Net"])
    ap.add_argument("--weights", default="")
    ap.add_argument("--align-corners", action="store_true",
                    help="bilinear upsampling as PyTorch 0.2 did it: what the published checkpoints were trained with")
    ap.add_argument("--save-which", type=int, default=1, choices=[0, 1], help="0: blended, 1: rectified (the demos' save_which)")
    a = ap.parse_args(argv)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP operators have no CPU fallback")
    import networks
    dev = torch.device("cuda", 0)
    net = getattr(networks, a.model)(channel=3, filter_size=4, training=False, align_corners=a.align_corners)
    if a.weights:
        # as the reference demo does (demo_MiddleBury.py:48-60): keep the checkpoint's entries that exist in the model, load
        # those, leave the rest of the model as constructed -- and say what was left out on either side
        state = torch.load(a.weights, map_location="cpu")
        state = state.get("state_dict", state)
        own = net.state_dict()
        kept = {k: v for k, v in state.items() if k in own}
        skipped, missing = sorted(set(state) - set(own)), sorted(set(own) - set(state))
        own.update(kept)
        net.load_state_dict(own, strict=True)
        if skipped or missing:
            print("checkpoint: %d entries loaded, %d not in the model (%s...), %d of the model not in the checkpoint (%s...)" % (
                len(kept), len(skipped), ", ".join(skipped[:3]), len(missing), ", ".join(missing[:3])))
    net = net.to(dev).eval()
    os.makedirs(a.output, exist_ok=True)
    results = networks.interpolate_png_tree(net, a.data, a.output, dev, gt_dir=a.gt or None, which=a.save_which)
    scored = [r for r in results if r[1] is not None]
    for scene, err, psnr in results:
        if err is None:
            print("%-16s written (no ground truth)" % scene)
        else:
            print("%-16s interpolation error / PSNR : %.4f / %.4f" % (scene, err, psnr))
    if scored:
        print("The average interpolation error / PSNR for all %d images are : %.4f / %.4f" % (
            len(scored), sum(r[1] for r in scored) / len(scored), sum(r[2] for r in scored) / len(scored)))


if __name__ == "__main__":
    main()
