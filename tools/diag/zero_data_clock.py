"""Is the conv stack held back by the chip's power management?  The same kernel on random data and on all-zero data
(zero weights and zero points: the same instruction stream, far fewer toggling bits)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import _lib, synth
from pointnetgpd_amd.pointnet import PointNetCls
B, P = 4096, 750
for prec in ("bf16", "bf16x3"):
    for zero in (False, True, False, True):
        m = PointNetCls(P, 3, 2)
        sd = synth.torch_state_dict(0, 2, 0.02)
        if zero:
            sd = {k: (torch.zeros_like(v) if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in sd.items()}
        m.load_state_dict(sd); m = m.cuda().eval(); m.precision = prec
        x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
        if zero:
            x.zero_()
        for _ in range(80): m(x)
        torch.cuda.synchronize()
        h = m._handle(x.device); lib = _lib.load()
        logp = torch.empty(B, 2, device="cuda"); trans = torch.empty(B, 3, 3, device="cuda"); ws = h.workspace(B)
        ms = (ctypes.c_float * 2)(); tot = 0.0; n = 40
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(n):
            _lib.check(lib.pngpd_forward_infer_timed(h.ptr, x.data_ptr(), 0, B, P, _lib.PRECISIONS[prec], logp.data_ptr(), trans.data_ptr(), None,
                                                     ws.data_ptr(), ws.numel(), ms, s))
            tot += ms[0] + ms[1]
        print(f"{prec} {'zero data  ' if zero else 'random data'}: conv stack {tot / (2 * n):.4f} ms", flush=True)
