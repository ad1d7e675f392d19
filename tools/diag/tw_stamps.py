"""Diagnostic: shader cycles per tile step of the training step's conv3 product/max kernel (workgroup 0).

    python tools/diag/stamps.py --build     (CPU box: builds tools/diag/libpngpd_diag.so with -DPNGPD_STAMPS)
    python tools/diag/tw_stamps.py          (GPU box)
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
from pointnetgpd_amd import training  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402

B, P, k = 256, 750, 2
m = PointNetCls(P, 3, k)
m.load_state_dict(synth.torch_state_dict(0, k, 0.02, 1.0))
m = m.cuda().train()
x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
ws = None
for _ in range(30):
    _, _, ws, _ = training.train_forward(m, x, ws)
torch.cuda.synchronize()
lib = _lib.load()
waves, nb, ne = 8, 6, 11
stamps = (ctypes.c_ulonglong * (waves * nb * ne))()
assert lib.pngpd_debug_read_tw_stamps(stamps) == 0
s = np.array(stamps, dtype=np.int64).reshape(waves, nb, ne)
np.set_printoptions(linewidth=250)
print("mean over batches 1..4, cycles: tiles 0..7 | wait at the barrier | behind it to the next batch start")
for w in range(waves):
    d = np.diff(s[w, 1:5, :10], axis=1).mean(axis=0)
    nxt = (s[w, 2:6, 0] - s[w, 1:5, 9]).mean()
    print(f"wave {w}: " + " ".join(f"{int(v):5d}" for v in d[:8]) + f" | {int(d[8]):5d} | {int(nxt):5d} | batch {int((s[w, 2:6, 0] - s[w, 1:5, 0]).mean()):6d}")
