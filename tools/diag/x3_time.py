"""Conv-stack time of the bf16x3 mode for the library given by --lib (tools/diag/<lib>.so), B=4096, P=750."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import _lib, synth
name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16x3"
if name:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", name + ".so")
from pointnetgpd_amd.pointnet import PointNetCls
B, P = 4096, 750
m = PointNetCls(P, 3, 2); m.load_state_dict(synth.torch_state_dict(0, 2, 0.02)); m = m.cuda().eval(); m.precision = prec
x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
for _ in range(60): m(x)
torch.cuda.synchronize()
h = m._handle(x.device); lib = _lib.load()
logp = torch.empty(B, 2, device="cuda"); trans = torch.empty(B, 3, 3, device="cuda"); ws = h.workspace(B)
ms = (ctypes.c_float * 2)(); tot = [0.0, 0.0]; n = 30
s = torch.cuda.current_stream().cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(n):
    _lib.check(lib.pngpd_forward_infer_timed(h.ptr, x.data_ptr(), 0, B, P, _lib.PRECISIONS[prec], logp.data_ptr(), trans.data_ptr(), None,
                                             ws.data_ptr(), ws.numel(), ms, s))
    tot[0] += ms[0]; tot[1] += ms[1]
ev1.record(); torch.cuda.synchronize()
print(f"{name or 'product'} {prec}: stn {tot[0]/n:.4f} ms  trunk {tot[1]/n:.4f} ms  step {ev0.elapsed_time(ev1)/n:.4f} ms", flush=True)
