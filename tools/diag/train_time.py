"""Training-step time (B = 256, P = 750) for the library tools/diag/<--lib>.so."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import _lib, synth
name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
if name:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", name + ".so")
from pointnetgpd_amd.pointnet import PointNetCls
from pointnetgpd_amd.training import Trainer
B, P, k = 256, 750, 2
m = PointNetCls(P, 3, k); m.load_state_dict(synth.torch_state_dict(0, k, 0.02, 1.0)); m = m.cuda().train()
tr = Trainer(m, lr=0.005)
x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
y = (torch.rand(B, device="cuda") < 0.5).long()
for _ in range(30): tr.step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): tr.step(x, y)
torch.cuda.synchronize()
print(f"{name or 'product'}: training step {1e3 * (time.perf_counter() - t0) / 100:.4f} ms", flush=True)
