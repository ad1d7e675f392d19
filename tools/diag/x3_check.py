"""Quick look at the bf16x3 mode: error against the golden files and the oracle, kernel time at B=4096."""
import glob, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls

G = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
def model(P, k, seed, scale, prec, gain=2.0):
    m = PointNetCls(P, 3, k); m.load_state_dict(synth.torch_state_dict(seed, k, scale, gain)); m = m.cuda().eval(); m.precision = prec
    return m
for path in sorted(glob.glob(os.path.join(G, "eval_*.npz"))):
    d = np.load(path)
    P, k, seed, B = (int(v) for v in d["meta"])
    x = torch.from_numpy(synth.candidate_points(seed, B, P, str(d["kind"]))).cuda()
    for prec in ("fp32", "bf16x3", "bf16"):
        m = model(P, k, seed, float(d["input_scale"]), prec, float(d["gain"]))
        logp, trans = m(x)
        print(os.path.basename(path), prec, "dlogp %.3e dtrans %.3e" % (np.abs(logp.cpu().numpy() - d["logp"]).max(), np.abs(trans.cpu().numpy() - d["trans"]).max()), flush=True)
# gain sweep
P, k, seed, B = 750, 2, 47, 64
x = torch.from_numpy(synth.candidate_points(seed, B, P, "gripper"))
for gain in (1.5, 2.5, 3.5, 4.5):
    sd = synth.torch_state_dict(seed, k, 0.02, gain)
    with torch.no_grad():
        ref = po.build_cls(P, k, sd)(x)[0].double()
    spread = (ref.max(1).values - ref.min(1).values).max().item()
    errs = {p: (model(P, k, seed, 0.02, p, gain)(x.cuda())[0].cpu().double() - ref).abs().max().item() for p in ("fp32", "bf16x3", "bf16")}
    print("gain", gain, "spread %.2f" % spread, {p: "%.2e" % e for p, e in errs.items()}, flush=True)
# timing
B = 4096
x = torch.rand(B, 3, 750, device="cuda") * 0.05
for prec in ("bf16", "bf16x3"):
    m = model(750, 2, 1, 0.02, prec)
    for _ in range(30): m(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(20): m(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 20
    print(prec, "B=4096: %.3f ms/step, %.3f M cand/s" % (dt * 1e3, B / dt / 1e6), flush=True)
# batch invariance
m = model(750, 2, 41, 0.02, "bf16x3")
x = torch.from_numpy(synth.candidate_points(41, 600, 750, "gripper")).cuda()
a, ta = m(x)
idx = torch.tensor([0, 1, 255, 256, 300, 599], device="cuda")
b, tb = m(x[idx].contiguous())
print("batch invariance:", torch.equal(a[idx], b), torch.equal(ta[idx], tb))
