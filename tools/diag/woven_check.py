"""Diagnostic: the woven and the phase form of the bf16 conv stack return identical bits.

    python tools/diag/woven_check.py        (GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointnetgpd_amd import synth  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ok = True
    for P, B in ((750, 4096), (500, 600), (1000, 300), (2500, 200), (752, 513), (241, 400), (256, 300), (737, 129)):
        model = PointNetCls(P, 3, 2)
        model.load_state_dict(synth.torch_state_dict(1, 2, 0.02), strict=True)
        model = model.to(dev).eval()
        model.precision = "bf16"
        x = torch.from_numpy(synth.candidate_points(P, B, P)).to(dev)
        outs = {}
        for form in ("phase", "woven"):
            os.environ["PNGPD_CONV_STACK_BF16"] = form
            logp, trans = model(x)
            torch.cuda.synchronize()
            outs[form] = (logp.cpu().numpy().copy(), trans.cpu().numpy().copy())
        same = all(np.array_equal(outs["phase"][i].view(np.uint32), outs["woven"][i].view(np.uint32)) for i in range(2))
        diff = float(np.abs(outs["phase"][0] - outs["woven"][0]).max())
        print(f"P={P} B={B}: identical={same} max|dlogp|={diff:.3e} finite={np.isfinite(outs['woven'][0]).all()}", flush=True)
        ok &= same
    print("ALL IDENTICAL" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
