"""Time of a weight reload (what the first eval forward after a training step pays): parameters on the GPU folded and
packed by kernels, against the host loader (PNGPD_HOST_FOLD=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointnetgpd_amd import synth                                   # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls, invalidate_packed_weights   # noqa: E402

P, k = 750, 2
m = PointNetCls(P, 3, k)
m.load_state_dict(synth.torch_state_dict(3, k, 1.0, 2.0))
m = m.cuda().eval()
x = torch.from_numpy(synth.candidate_points(3, 1, P)).cuda()
for how in ("device", "host", "device", "host"):
    if how == "host":
        os.environ["PNGPD_HOST_FOLD"] = "1"
    else:
        os.environ.pop("PNGPD_HOST_FOLD", None)
    m(x)
    torch.cuda.synchronize()
    t = []
    for _ in range(20):
        invalidate_packed_weights()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t.append((t1 - t0, t2 - t0))
    t.sort(key=lambda v: v[1])
    print(f"{how}: reload + one forward, median {t[10][1] * 1e6:.0f} us until the result is there "
          f"({t[10][0] * 1e6:.0f} us until the call returns)")
