"""Per-call latency of the batch-1 scoring shim and of small eval batches (wall clock, device synchronised)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnetgpd_amd import synth  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402
from pointnetgpd_amd.scoring import test_network  # noqa: E402

P, k = 750, 2
m = PointNetCls(P, 3, k)
m.load_state_dict(synth.torch_state_dict(0, k, 0.02))
m = m.cuda().eval()
for prec in ("bf16", "bf16x3", "fp32"):
    m.precision = prec
    pc = synth.candidate_points(0, 1, P)[0].T.astype(np.float64).copy()       # [P,3] like the reference's local_pc
    for _ in range(20):
        test_network(m, pc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        test_network(m, pc)
    torch.cuda.synchronize()
    print(f"{prec}: test_network batch-1: {1e6 * (time.perf_counter() - t0) / n:.1f} us per call")
    for B in (1, 3, 16, 64, 256, 1024):
        x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{prec}: forward B={B}: {1e6 * dt:.1f} us per call, {B / dt:,.0f} candidates/s")
