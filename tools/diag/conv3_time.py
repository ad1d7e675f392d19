"""Duration of the training step's conv3 product/max kernel (HIP events around it), B = 256, P = 750, for the library
tools/diag/<--lib>.so (default: the product library) and the mode in PNGPD_TRAIN_CONV3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import _lib, synth
name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
if name:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", name + ".so")
from pointnetgpd_amd import training
from pointnetgpd_amd.pointnet import PointNetCls
B, P, k = 256, 750, 2
m = PointNetCls(P, 3, k); m.load_state_dict(synth.torch_state_dict(0, k, 0.02, 1.0)); m = m.cuda().train()
x = torch.from_numpy(synth.candidate_points(0, B, P)).cuda()
ws = None
for _ in range(20):
    _, _, ws, _ = training.train_forward(m, x, ws)
ms = []
for _ in range(30):
    _, _, ws, _ = training.train_forward(m, x, ws, conv3_ms=ms)
torch.cuda.synchronize()
print(f"{name or 'product'} mode={os.environ.get('PNGPD_TRAIN_CONV3', 'default')}: conv3 product/max kernel {1e3 * sum(ms) / len(ms):.1f} us "
      f"(min {1e3 * min(ms):.1f})", flush=True)
