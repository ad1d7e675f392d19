"""Randomised size sweep of the forward path against the CPU oracle (run by hand on a GPU box:
    python tests/sweep_forward.py); not collected by pytest."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import pointnet_oracle as po
from pointnetgpd_amd import synth
from pointnetgpd_amd.pointnet import PointNetCls
rng = np.random.default_rng(123)
worst = {"fp32": 0.0, "bf16": 0.0}
for it in range(40):
    P = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 47, 48, 100, 240, 241, 255, 256, 257, 272, 400, 511, 512, 513, 750, 769, 1000, 1025]))
    B = int(rng.choice([1, 2, 7, 64, 255, 256, 257, 300, 511, 513]))
    if B * P > 300 * 1100: B = max(1, 300 * 1100 // P)
    k = int(rng.choice([2, 3]))
    seed = 1000 + it
    sd = synth.torch_state_dict(seed, k, 0.02, 1.0)     # gain 1: |logit| of a few units, as trained checkpoints give
    oracle = po.build_cls(P, k, sd)
    x = torch.from_numpy(synth.candidate_points(seed, B, P, "gripper"))
    with torch.no_grad():
        ref, rtrans = oracle(x)
    for prec, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        m = PointNetCls(P, 3, k); m.load_state_dict(sd); m = m.cuda().eval(); m.precision = prec
        out, trans = m(x.cuda())
        err = (out.cpu() - ref).abs().max().item()
        worst[prec] = max(worst[prec], err)
        status = "ok" if err <= tol else "FAIL"
        print(f"{it:2d} B={B:4d} P={P:5d} k={k} {prec}: {err:.2e} (max |logp| {ref.abs().max().item():.1f}) {status}", flush=True)
        assert err <= tol
print("worst", worst)
