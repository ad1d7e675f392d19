"""Diagnostic: shader cycles per tile step of the woven bf16 conv-stack kernel (workgroup 0).

    python tools/diag/stamps.py --build     (CPU box: builds tools/diag/libpngpd_diag.so with -DPNGPD_STAMPS)
    python tools/diag/wstamps.py            (GPU box)
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pointnetgpd_amd import _lib, synth  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", "libpngpd_diag.so")
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402


def main(B=4096, P=750):
    dev = torch.device("cuda", 0)
    model = PointNetCls(P, 3, 2)
    model.load_state_dict(synth.torch_state_dict(0, 2, 0.02), strict=True)
    model = model.to(dev).eval()
    model.precision = "bf16"
    x = torch.from_numpy(synth.candidate_points(0, B, P)).to(dev)
    for _ in range(300):
        model(x)
    torch.cuda.synchronize()
    lib = _lib.load()
    waves, nb, ne = 8, 6, 18
    stamps = (ctypes.c_ulonglong * (waves * nb * ne))()
    assert lib.pngpd_debug_read_wstamps(stamps) == 0
    s = np.array(stamps, dtype=np.int64).reshape(waves, nb, ne)
    d = np.diff(s, axis=2)          # [wave][batch][17]: steps 0..15, barrier wait
    np.set_printoptions(linewidth=250)
    print("mean over batches, cycles per tile step 0..15 | barrier wait | total")
    for w in range(waves):
        m = d[w].mean(axis=0)
        print(f"wave {w}: " + " ".join(f"{int(v):5d}" for v in m[:16]) + f" | {int(m[16]):5d} | {int((s[w, :, 17] - s[w, :, 0]).mean()):6d}")
    print("batch starts relative to wave 0, batch 0:")
    print(s[:, :, 0] - s[0, 0, 0])


if __name__ == "__main__":
    main()
