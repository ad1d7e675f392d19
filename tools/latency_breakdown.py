"""Where a small-batch forward spends its time: Python/ctypes host overhead per call (no synchronisation between
calls, so the stream never drains), GPU time per call (HIP events), and wall clock per synchronised call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnetgpd_amd import synth  # noqa: E402
from pointnetgpd_amd.pointnet import PointNetCls  # noqa: E402

P, k = 750, 2
m = PointNetCls(P, 3, k)
m.load_state_dict(synth.torch_state_dict(0, k, 0.02))
m = m.cuda().eval()
for prec in ("bf16", "fp32"):
    m.precision = prec
    for B in (1, 16, 64):
        x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
        for _ in range(50):
            m(x)
        torch.cuda.synchronize()
        n = 300
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            m(x)
        e1.record()
        host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        gpu = e0.elapsed_time(e1) * 1e-3 / n
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
            torch.cuda.synchronize()
        sync = (time.perf_counter() - t0) / n
        h = m._handle(x.device)
        t0 = time.perf_counter()
        for _ in range(n):
            m._handle(x.device)
        sig = (time.perf_counter() - t0) / n
        print(f"{prec} B={B}: host issue {1e6 * host:.1f} us/call, stream time {1e6 * gpu:.1f} us/call, "
              f"synchronised call {1e6 * sync:.1f} us, of which the weight-change check {1e6 * sig:.1f} us")
