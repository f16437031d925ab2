"""Records the outputs of the reference's own GPU kernels (oracle/ref_gpu.py: my_lib_kernel.cu built for gfx950 from
the reference tree) on seeded inputs, as fixtures tests/golden/ref_gpu_<case>.npz -- inputs are regenerated from the
seed (tests/_refcases.py), only outputs are stored.  Run ON AN MI355X (through gpurun):

    python tests/golden/make_golden_ref_gpu.py gpurun_out/ref_golden      # then copy *.npz into tests/golden/

tests/test_golden_reference.py holds the CPU oracle to these vectors -- on any machine, no GPU needed."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(HERE))
import _refcases as RC                     # noqa: E402
from oracle import ref_gpu as R            # noqa: E402


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_golden")
    os.makedirs(outdir, exist_ok=True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    N = lambda t: t.detach().cpu().numpy()                               # noqa: E731
    for case in RC.REF_CASES[:3] + RC.REF_CASES[3:]:
        out = RC.reference_outputs(R, RC.make(case), T, N)
        out["device"] = np.array(torch.cuda.get_device_name(0))
        np.savez_compressed(os.path.join(outdir, "ref_gpu_%s.npz" % RC.name(case)), **out)
        print(RC.name(case), {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    main()
