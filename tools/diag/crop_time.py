"""Time of the crop + subsample stage (S = 1M, G = 10k, R = 3, P = 750) for the library tools/diag/<--lib>.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pointnetgpd_amd import _lib, synth
name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else None
if name:
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", name + ".so")
from pointnetgpd_amd import crop as hc
S, G, P, R = 1_000_000, 10_000, 750, 3
limits = hc.box_limits_from_gripper(synth.GRIPPER_BOX[0], synth.GRIPPER_BOX[1] + 0.04, 0.02, synth.GRIPPER_BOX[2])
scene = torch.from_numpy(synth.scene_points(0, S, 0.5)).cuda()
grasps = torch.from_numpy(synth.random_grasps(1000, G, 0.5, 0.06)).cuda()
for _ in range(10):
    hc.crop_subsample(grasps, scene, limits, P, repeat=R, seed=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    hc.crop_subsample(grasps, scene, limits, P, repeat=R, seed=0)
e1.record(); torch.cuda.synchronize()
print(f"{name or 'product'}: crop + subsample stage {e0.elapsed_time(e1) / 30:.4f} ms", flush=True)
